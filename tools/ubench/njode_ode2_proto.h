// njode_ode2_proto.h -- measured-and-rejected prototypes of the segment plan's ODE backward
// (DESIGN.md section 4a''), kept for the micro-benchmark tools/ubench/ode_ubench.hip only; no
// product translation unit includes this file.
//   * ode2_bwd_single: the one-wave backward that RECOMPUTES the hidden layers, on the scaled
//     fragments (superseded by ode3_bwd_single on stored activations, njode_ode2.h);
//   * ode3_bwd_pair (round 3): two tiles per wave at one wave per SIMD, 0.60-0.66 ms against
//     0.50 ms for two waves per SIMD with one tile each (profiles/r03_bwd_experiments.jsonl);
//   * k_ode2_bwd_pc: producer / consumer split of the sweep inside a 512-thread block (correct,
//     no gain: f32 MFMA and VALU share one pipe, both halves sit on it).
#pragma once
#include "../../njode_amd/csrc/njode_ode2.h"

namespace njode {

// ---- the one-wave backward of njode_mfma.h on the scaled fragments ---------------------------
// hidden activation from accumulator tiles of PRE-SCALED pre-activations: a[q] = act (+ dropout
// select, no scale), bias unit = 1
template <class C, bool DROP>
NJ_DEV void hidden_from_acc2(const f32x4 (&acc)[MF<C>::MT1], float (&av)[MF<C>::Q1], uint32_t keep, int g) {
  constexpr int Q1 = MF<C>::Q1;
#pragma unroll
  for (int q = 0; q < Q1; ++q) {
    float v = act2_f<C::ACT>(acc[q / 4][q % 4]);
    if constexpr (DROP) v = ((keep >> q) & 1) ? v : 0.0f;
    av[q] = v;
  }
  constexpr int QB = MF<C>::W / 4, GB = MF<C>::W % 4;
  av[QB] = g == GB ? 1.0f : av[QB];
}
// delta of a hidden layer: the transposed product already carries 1 / (1 - p) (fragments B3 / B2)
template <class C, bool DROP>
NJ_DEV void hidden_delta2(const f32x4 (&acc)[MF<C>::MT1], const float (&av)[MF<C>::Q1],
                          float (&dv)[MF<C>::QW], uint32_t keep) {
#pragma unroll
  for (int q = 0; q < MF<C>::QW; ++q) {
    const float d = acc[q / 4][q % 4] * dact_f<C::ACT>(av[q]);
    if constexpr (DROP) dv[q] = ((keep >> q) & 1) ? d : 0.0f;
    else dv[q] = d;
  }
}

template <class C, bool DROP>
NJ_DEV void ode2_bwd_single(const KArgs& a, lfp lds_raw, int wave, int n_waves, int tile0, int tile1,
                           int slab_row) {
  using M = MF<C>;
  using NL = typename C::Ode;
  using FR = OdeLdsFrags<C>;
  constexpr int NT1 = (M::W + 1 + 15) / 16;     // column tiles of [a, 1]
  constexpr int NT0 = (M::IN0 + 1 + 15) / 16;   // column tiles of [in0, 1]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  lfp img_d = lds_raw + wv * 2 * IMG_FLOATS, img_a = img_d + IMG_FLOATS;
  lfp fimg = lds_raw + 4 * 2 * IMG_FLOATS;
  FR::stage(fimg, a.frag2, threadIdx.x, 256);   // scaled fragments (k_pack_frags2)
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
      hidden_from_acc2<C, DROP>(acc, a1, k1, g);
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = zero4;
#pragma unroll
      for (int q = 0; q < M::Q1; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(F.a2(mt, q), a1[q], acc[mt]);
      hidden_from_acc2<C, DROP>(acc, a2, k2, g);

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
      hidden_delta2<C, DROP>(acc, a2, d2, k2);
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
      hidden_delta2<C, DROP>(acc, a1, d1, k1);
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
  // the activations carry no inverted-dropout factor here: it goes on once, at the flush
  const float ik = DROP ? a.dc.inv_keep : 1.0f;
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
          if (ui < M::W) W2[uo * M::W + ui] = ik * G2[mt][nt][r];
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
          if (ui < M::W) W3[uo * M::W + ui] = ik * G3[mt][nt][r];
          else if (ui == M::W) b3[uo] = G3[mt][nt][r];
        }
      }
    }
}



// ---- backward: producer / consumer ------------------------------------------------------
// One image set = the six [unit][chain] images of one Euler step of one tile, rows packed
// back to back (row stride IMG_STRIDE floats).  The consumer reads whole 16-row tiles, so a
// read of an image's last tile may run into the next image: those rows only meet
// accumulator entries that are never flushed.
template <class C> struct Ode2Img {
  using M = MF<C>;
  static constexpr int NT1 = (M::W + 1 + 15) / 16, NT0 = (M::IN0 + 1 + 15) / 16;
  static constexpr int D3 = 0;
  static constexpr int A2 = D3 + 4 * M::QH;
  static constexpr int D2 = A2 + 4 * M::Q1;
  static constexpr int A1 = D2 + 4 * M::QW;
  static constexpr int D1 = A1 + 4 * M::Q1;
  static constexpr int B0 = D1 + 4 * M::QW;
  static constexpr int ROWS_USED = B0 + 4 * M::Q0;
  // the last tile read of every image must stay inside the set
  static constexpr int cmax(int a, int b) { return a > b ? a : b; }
  static constexpr int ROWS =
      cmax(cmax(cmax(D3 + 16 * M::MTH, A2 + 16 * NT1), cmax(D2 + 16 * M::MT1, A1 + 16 * NT1)),
           cmax(cmax(D1 + 16 * M::MT1, B0 + 16 * NT0), ROWS_USED));
  static constexpr int SET_FLOATS = ROWS * IMG_STRIDE;
  static constexpr int NG = M::MTH * NT1 + M::MT1 * NT1 + M::MT1 * NT0;
  // 4 pairs x 2 sets, + round bookkeeping; the final tile reduction reuses the sets
  static constexpr int RED_FLOATS = 3 * NG * 64 * 4;
  static constexpr int IMG_FLOATS_ALL = 4 * 2 * SET_FLOATS;
  static constexpr int FLOATS = (IMG_FLOATS_ALL > RED_FLOATS ? IMG_FLOATS_ALL : RED_FLOATS) + 16;
};

template <class C> struct Ode2BwdFrags {
  using M = MF<C>;
  float A1[M::MT1][M::Q0], A2[M::MT1][M::Q1];
  float B3[M::MT1][M::QH], B2[M::MT1][M::QW], B1[M::MTH][M::QW];
  NJ_DEV void load(const float* frag, int lane) {
#pragma unroll
    for (int mt = 0; mt < M::MT1; ++mt) {
#pragma unroll
      for (int q = 0; q < M::Q0; ++q) A1[mt][q] = frag[(M::F1 + mt * M::Q0 + q) * 64 + lane];
#pragma unroll
      for (int q = 0; q < M::Q1; ++q) A2[mt][q] = frag[(M::F2 + mt * M::Q1 + q) * 64 + lane];
#pragma unroll
      for (int q = 0; q < M::QH; ++q) B3[mt][q] = frag[(M::B3 + mt * M::QH + q) * 64 + lane];
#pragma unroll
      for (int q = 0; q < M::QW; ++q) B2[mt][q] = frag[(M::B2 + mt * M::QW + q) * 64 + lane];
    }
#pragma unroll
    for (int mt = 0; mt < M::MTH; ++mt)
#pragma unroll
      for (int q = 0; q < M::QW; ++q) B1[mt][q] = frag[(M::B1 + mt * M::QW + q) * 64 + lane];
  }
};

// Recompute of one hidden layer for the sweep: a[q] = kept ? act(z) : 0 (written to the
// image `img_a` for the dW product), da[q] = kept ? act'(z) : 0 (kept for the delta).
template <class C, bool DROP, int MT0>
NJ_DEV void hidden_pair_bwd(const f32x4& t0, const f32x4& t1, float (&av)[MF<C>::Q1],
                            float (&da)[MF<C>::Q1], uint32_t& s, uint32_t thr16, int g) {
  constexpr int Q1 = MF<C>::Q1;
#pragma unroll
  for (int r = 0; r < 8; r += 2) {
    const int q = 4 * MT0 + r;
    if (q < Q1) {
      float v0 = act2_f<C::ACT>(r < 4 ? t0[r & 3] : t1[r & 3]);
      float v1 = q + 1 < Q1 ? act2_f<C::ACT>(r + 1 < 4 ? t0[(r + 1) & 3] : t1[(r + 1) & 3]) : 0.0f;
      float d0 = dact_f<C::ACT>(v0), d1 = dact_f<C::ACT>(v1);
      if constexpr (DROP) {
        const uint32_t w = xs32(s);
        const bool k0 = (w & 0xffffu) >= thr16, k1 = (w >> 16) >= thr16;
        v0 = k0 ? v0 : 0.0f;
        d0 = k0 ? d0 : 0.0f;
        v1 = k1 ? v1 : 0.0f;
        d1 = k1 ? d1 : 0.0f;
      }
      av[q] = v0;
      da[q] = d0;
      if (q + 1 < Q1) { av[q + 1] = v1; da[q + 1] = d1; }
    }
  }
  constexpr int QB = MF<C>::W / 4, GB = MF<C>::W % 4;
  if constexpr (QB >= 4 * MT0 && QB < 4 * MT0 + 8) {
    av[QB] = g == GB ? 1.0f : av[QB];   // bias unit
    da[QB] = g == GB ? 0.0f : da[QB];
  }
}
template <class C, bool DROP, int QIN, int MT0 = 0>
NJ_DEV void hidden_layer2_bwd(const float (&A)[MF<C>::MT1][QIN], const float (&bv)[QIN],
                              float (&av)[MF<C>::Q1], float (&da)[MF<C>::Q1], uint32_t& st,
                              uint32_t thr16, int g) {
  if constexpr (MT0 < MF<C>::MT1) {
    f32x4 t0, t1;
    mfma_pair<MF<C>::MT1, QIN, MT0>(A, bv, t0, t1);
    hidden_pair_bwd<C, DROP, MT0>(t0, t1, av, da, st, thr16, g);
    hidden_layer2_bwd<C, DROP, QIN, MT0 + 2>(A, bv, av, da, st, thr16, g);
  }
}
// dv[q] = (B x din)[unit 4q + g] * da[q] for the W hidden units, output tiles in pairs
template <class C, int QIN, int MT0 = 0>
NJ_DEV void delta_layer2(const float (&Bf)[MF<C>::MT1][QIN], const float (&din)[QIN],
                         const float (&da)[MF<C>::Q1], float (&dv)[MF<C>::QW]) {
  if constexpr (MT0 < MF<C>::MT1) {
    f32x4 t0, t1;
    mfma_pair<MF<C>::MT1, QIN, MT0>(Bf, din, t0, t1);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int q = 4 * MT0 + r;
      if (q < MF<C>::QW) dv[q] = (r < 4 ? t0[r & 3] : t1[r & 3]) * da[q];
    }
    delta_layer2<C, QIN, MT0 + 2>(Bf, din, da, dv);
  }
}

NJ_DEV void pc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// producer: adjoint sweep of the tiles 4 * group + p, group walking the block's rounds
template <class C, bool DROP>
NJ_DEV void ode2_bwd_producer(const KArgs& a, lfp lds, int p, int lane, int n_tiles) {
  using M = MF<C>;
  using I = Ode2Img<C>;
  const int g = lane >> 4, c = lane & 15;
  Ode2BwdFrags<C> F;
  F.load(a.frag2, lane);
  lfp sets = lds + p * 2 * I::SET_FLOATS;
  int __attribute__((address_space(3)))* nm = (int __attribute__((address_space(3)))*)(lds + I::FLOATS - 16);
  const int n_groups = (n_tiles + 3) / 4;
  float* const trash = a.trash + (p * 64 + lane) * C::H;
  int gs = 0;   // global step counter: image set = gs & 1
  for (int round = 0; round * (int)gridDim.x < n_groups; ++round) {
    const int grp = snake_tile(round, blockIdx.x, gridDim.x);
    const int tile = 4 * grp + p;
    const bool tile_ok = grp < n_groups && tile < n_tiles;
    const int j = tile * 16 + c;
    const bool valid = tile_ok && j < a.n_obs;
    Item<C> it;
    it.template load<false>(a, j, valid);
    float lam[M::QH];
#pragma unroll
    for (int q = 0; q < M::QH; ++q) {
      const int u = 4 * q + g;
      const float v = a.lam_end[(size_t)it.r * C::H + (u < C::H ? u : 0)];
      lam[q] = (valid && u < C::H) ? v : 0.0f;
    }
    const int nmax_w = wave_max(it.n);
    if (lane == 0) nm[(round & 1) * 4 + p] = nmax_w;
    pc_barrier();                                        // round barrier
    const int nmax = max(max(nm[(round & 1) * 4 + 0], nm[(round & 1) * 4 + 1]),
                         max(nm[(round & 1) * 4 + 2], nm[(round & 1) * 4 + 3]));
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
    for (int s = nmax - 1; s >= 0; --s, ++gs) {
      lfp img = sets + (gs & 1) * I::SET_FLOATS;
      const bool active = s < it.n;
      const int k = active ? it.kbeg + s : 0;
      float h[M::QH];
#pragma unroll
      for (int q = 0; q < M::QH; ++q) h[q] = h_n[q];
      const float dt = dt_n, t = t_n;
      if (s > 0) fetch(s - 1, h_n, dt_n, t_n);
      float b0[M::Q0], d3[M::QH];
      in0_fill<C, 0>(b0, h, it.tx, it.tau, t - it.tau, g);
#pragma unroll
      for (int q = 0; q < M::QH; ++q) d3[q] = dt * lam[q];   // delta3 (zero for inactive chains)
      img_write<M::QH>(img + I::D3 * IMG_STRIDE, d3, g, c);
      img_write<M::Q0>(img + I::B0 * IMG_STRIDE, b0, g, c);
      uint32_t st = 0;
      if constexpr (DROP) {
        const unsigned long long gid = a.gid0 + it.b;
        st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1),
                        (uint32_t)k, NET_ODE);
      }
      // ---- recompute the two hidden layers
      float a1[M::Q1], da1[M::Q1], a2[M::Q1], da2[M::Q1];
      hidden_layer2_bwd<C, DROP, M::Q0>(F.A1, b0, a1, da1, st, a.dc.thr16, g);
      img_write<M::Q1>(img + I::A1 * IMG_STRIDE, a1, g, c);
      hidden_layer2_bwd<C, DROP, M::Q1>(F.A2, a1, a2, da2, st, a.dc.thr16, g);
      img_write<M::Q1>(img + I::A2 * IMG_STRIDE, a2, g, c);
      // ---- deltas
      float d2[M::QW], d1[M::QW];
      delta_layer2<C, M::QH>(F.B3, d3, da2, d2);
      img_write<M::QW>(img + I::D2 * IMG_STRIDE, d2, g, c);
      delta_layer2<C, M::QW>(F.B2, d2, da1, d1);
      img_write<M::QW>(img + I::D1 * IMG_STRIDE, d1, g, c);
      // ---- adjoint of the state: lam += (W1^T delta1)[h rows] * (1 - tanh(h)^2)
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < M::MTH; ++mt) {
        f32x4 e = z, o = z;
#pragma unroll
        for (int q = 0; q < M::QW; q += 2) {
          e = mfma4(F.B1[mt][q], d1[q], e);
          if (q + 1 < M::QW) o = mfma4(F.B1[mt][q + 1], d1[q + 1], o);
        }
        e += o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = 4 * mt + r;
          if (q < M::QH) {
            const float th = b0[q];  // = tanh(h) wherever unit 4q + g < H
            const float dth = (4 * q + g) < C::H ? 1.0f - th * th : 0.0f;
            lam[q] = fmaf(e[r], dth, lam[q]);
          }
        }
      }
      pc_barrier();                                      // step barrier: set gs & 1 is complete
    }
    float* out = valid ? a.lam_start + (size_t)it.r * C::H : trash;
#pragma unroll
    for (int q = 0; q < M::QH; ++q) {
      const int u = 4 * q + g;
      float* dst = u < C::H ? out + u : trash;
      *dst = lam[q];
    }
  }
}

// consumer: dW accumulator tiles; reads the partner's image sets
template <class C>
NJ_DEV void ode2_bwd_consumer(const KArgs& a, lfp lds, int p, int lane, int n_tiles, int slab_row,
                              float ik) {
  using M = MF<C>;
  using I = Ode2Img<C>;
  using NL = typename C::Ode;
  constexpr int NT1 = I::NT1, NT0 = I::NT0;
  const int g = lane >> 4, c = lane & 15;
  lfp sets = lds + p * 2 * I::SET_FLOATS;
  int __attribute__((address_space(3)))* nm = (int __attribute__((address_space(3)))*)(lds + I::FLOATS - 16);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 G3[M::MTH][NT1], G2[M::MT1][NT1], G1[M::MT1][NT0];
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
  const int n_groups = (n_tiles + 3) / 4;
  int gs = 0;
  for (int round = 0; round * (int)gridDim.x < n_groups; ++round) {
    pc_barrier();                                        // round barrier
    const int nmax = max(max(nm[(round & 1) * 4 + 0], nm[(round & 1) * 4 + 1]),
                         max(nm[(round & 1) * 4 + 2], nm[(round & 1) * 4 + 3]));
    for (int s = nmax - 1; s >= 0; --s, ++gs) {
      pc_barrier();                                      // step barrier: set gs & 1 is complete
      lfp img = sets + (gs & 1) * I::SET_FLOATS;
      dw_accumulate<M::MTH, NT1>(img + I::D3 * IMG_STRIDE, img + I::A2 * IMG_STRIDE, G3, g, c);
      dw_accumulate<M::MT1, NT1>(img + I::D2 * IMG_STRIDE, img + I::A1 * IMG_STRIDE, G2, g, c);
      dw_accumulate<M::MT1, NT0>(img + I::D1 * IMG_STRIDE, img + I::B0 * IMG_STRIDE, G1, g, c);
    }
  }
  // ---- flush: the four consumers sum their tiles through LDS (fixed order) into ONE slab row
  constexpr int NG = I::NG;
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
  f32x4 __attribute__((address_space(3)))* red = (f32x4 __attribute__((address_space(3)))*)lds;
  pc_barrier();                                          // every image read is done
  if (p > 0) for_tiles([&](f32x4& t, int i) { red[((p - 1) * NG + i) * 64 + lane] = t; });
  pc_barrier();
  if (p != 0) return;
  for_tiles([&](f32x4& t, int i) {
    t += red[(0 * NG + i) * 64 + lane];
    t += red[(1 * NG + i) * 64 + lane];
    t += red[(2 * NG + i) * 64 + lane];
  });
  float* slab = a.slab + (size_t)slab_row * C::P + C::OFF_ODE;
  float *W1 = slab + NL::woff(0), *b1 = slab + NL::boff(0), *W2 = slab + NL::woff(1),
        *b2 = slab + NL::boff(1), *W3 = slab + NL::woff(2), *b3 = slab + NL::boff(2);
  // the activations in the images carry no inverted-dropout factor: it goes on here
#pragma unroll
  for (int mt = 0; mt < M::MT1; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int uo = 16 * mt + 4 * g + r;
      if (uo < M::W) {
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) {
          const int ui = 16 * nt + c;
          if (ui < M::W) W2[uo * M::W + ui] = ik * G2[mt][nt][r];
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
          if (ui < M::W) W3[uo * M::W + ui] = ik * G3[mt][nt][r];
          else if (ui == M::W) b3[uo] = G3[mt][nt][r];
        }
      }
    }
}

// C (v2): reverse Euler sweep + d loss / d ODE params; one slab row per block (row blockIdx.x)
template <class C, bool DROP>
__global__ void __launch_bounds__(512, 2) k_ode2_bwd_pc(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ode2_lds[];
  lfp lds = (lfp)ode2_lds;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n_tiles = (a.n_obs + 15) / 16;
  for (int i = threadIdx.x; i < Ode2Img<C>::FLOATS; i += 512) lds[i] = 0.0f;
  __syncthreads();
  if (wv < 4) {
    ode2_bwd_producer<C, DROP>(a, lds, wv, lane, n_tiles);
    pc_barrier();   // the consumers' two reduction barriers
    pc_barrier();
  } else {
    ode2_bwd_consumer<C>(a, lds, wv - 4, lane, n_tiles, blockIdx.x, DROP ? a.dc.inv_keep : 1.0f);
  }
}



// ---- C (stored activations), TWO tiles per wave (round 3) ------------------------------------
// ode3_bwd_single at two waves per SIMD leaves the shared f32 pipe ~1/3 idle: a lone wave keeps
// it ~60 % busy, and the partner's instructions come out of the same pipe.  Here
// ONE wave per SIMD (512 registers) carries two independent tiles through the sweep: the
// compiler interleaves the two dependency chains statically (a tile's LDS round trips and
// MFMA -> VALU hazards are covered by the other tile's MFMAs), every A-fragment read from LDS
// feeds two MFMAs, and both tiles accumulate into ONE set of dW tiles (K = 32 chains per step).
// Tiles 2p and 2p + 1 of the length-sorted order are paired: (almost) equal lengths.
template <class C> struct OdeBwdPairLds {
  using M = MF<C>;
  static constexpr int NG = OdeBwdActLds<C>::NG;
  static constexpr int BODY = 4 * 4 * IMG_FLOATS + OdeLdsFragsT<C>::NVEC * 64;
  static constexpr int RED = 3 * NG * 64 * 4;
  static constexpr int FLOATS = BODY > RED ? BODY : RED;
};
template <class C, bool DROP>
NJ_DEV void ode3_bwd_pair(const KArgs& a, lfp lds_raw, int wave, int n_waves, int tile0, int tile1,
                          int slab_row) {
  using M = MF<C>;
  using NL = typename C::Ode;
  using FR = OdeLdsFragsT<C>;
  constexpr int NT1 = (M::W + 1 + 15) / 16;     // column tiles of [a, 1]
  constexpr int NT0 = (M::IN0 + 1 + 15) / 16;   // column tiles of [in0, 1]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  lfp img_d[2], img_a[2];
  img_d[0] = lds_raw + wv * 4 * IMG_FLOATS;
  img_a[0] = img_d[0] + IMG_FLOATS;
  img_d[1] = img_a[0] + IMG_FLOATS;
  img_a[1] = img_d[1] + IMG_FLOATS;
  lfp fimg = lds_raw + 4 * 4 * IMG_FLOATS;
  FR::stage(fimg, a.frag2, threadIdx.x, 256);
  for (int i = threadIdx.x; i < 4 * 4 * IMG_FLOATS; i += 256) lds_raw[i] = 0.0f;
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
  const int n_pairs = (n_tiles + 1) / 2;
  for (int round = 0; round * n_waves < n_pairs; ++round) {
    const int rel = snake_tile(round, wave, n_waves);
    if (rel >= n_pairs) continue;
    int tile[2], j[2];
    bool valid[2];
    Item<C> it[2];
    float lam[2][M::QH];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int tl = tile0 + 2 * rel + t;
      tile[t] = tl < tile1 ? tl : tile1 - 1;          // odd tail: tile repeated with no valid item
      j[t] = tile[t] * 16 + c;
      valid[t] = tl < tile1 && j[t] < a.n_obs;
      it[t].template load<false>(a, j[t], valid[t]);
#pragma unroll
      for (int q = 0; q < M::QH; ++q) {
        const int u = 4 * q + g;
        const float v = a.lam_end[(size_t)it[t].r * C::H + (u < C::H ? u : 0)];
        lam[t][q] = (valid[t] && u < C::H) ? v : 0.0f;
      }
    }
    const int nmax = wave_max(it[0].n > it[1].n ? it[0].n : it[1].n);
    auto fetch = [&](int t, int s, float (&hh)[M::QH], float (&x1)[M::Q1], float (&x2)[M::Q1], float& dtt,
                     float& tt) {
      const bool act = s < it[t].n;
      const int kk = act ? it[t].kbeg + s : 0;
      const float* rec = a.traj + (act ? (size_t)(a.base_s[s] + j[t]) * C::H : 0);
#pragma unroll
      for (int q = 0; q < M::QH; ++q) {
        const int u = 4 * q + g;
        const float v = rec[u < C::H ? u : 0];
        hh[q] = u < C::H ? v : 0.0f;
      }
      act_load<C>(a.act, a.base16_s[s], tile[t], lane, x1, x2);
      dtt = act ? a.step_dt[kk] : 0.0f;
      tt = a.step_t[kk];
    };
    float h_n[2][M::QH], a1_n[2][M::Q1], a2_n[2][M::Q1], dt_n[2] = {0.0f, 0.0f}, t_n[2] = {0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int q = 0; q < M::QH; ++q) h_n[t][q] = 0.0f;
#pragma unroll
      for (int q = 0; q < M::Q1; ++q) { a1_n[t][q] = 0.0f; a2_n[t][q] = 0.0f; }
    }
    if (nmax > 0) {
      fetch(0, nmax - 1, h_n[0], a1_n[0], a2_n[0], dt_n[0], t_n[0]);
      fetch(1, nmax - 1, h_n[1], a1_n[1], a2_n[1], dt_n[1], t_n[1]);
    }
    for (int s = nmax - 1; s >= 0; --s) {
      float h[2][M::QH], a1[2][M::Q1], a2[2][M::Q1], dt[2], tm[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int q = 0; q < M::QH; ++q) h[t][q] = h_n[t][q];
#pragma unroll
        for (int q = 0; q < M::Q1; ++q) { a1[t][q] = a1_n[t][q]; a2[t][q] = a2_n[t][q]; }
        dt[t] = dt_n[t];
        tm[t] = t_n[t];
      }
      if (s > 0) {
        fetch(0, s - 1, h_n[0], a1_n[0], a2_n[0], dt_n[0], t_n[0]);
        fetch(1, s - 1, h_n[1], a1_n[1], a2_n[1], dt_n[1], t_n[1]);
      }
      float b0[2][M::Q0];
      in0_fill<C, 0>(b0[0], h[0], it[0].tx, it[0].tau, tm[0] - it[0].tau, g);
      in0_fill<C, 0>(b0[1], h[1], it[1].tx, it[1].tau, tm[1] - it[1].tau, g);
      F.begin();

      // ---- layer 3: delta3 = dt * lam (zero for inactive chains)
      float d3[2][M::QH];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int q = 0; q < M::QH; ++q) d3[t][q] = dt[t] * lam[t][q];
        img_write<M::QH>(img_d[t], d3[t], g, c);
        img_write<M::Q1>(img_a[t], a2[t], g, c);
      }
      wave_lds_sync();
      dw_accumulate<M::MTH, NT1>(img_d[0], img_a[0], G3, g, c);
      dw_accumulate<M::MTH, NT1>(img_d[1], img_a[1], G3, g, c);
      f32x4 acc[2][M::MT1];
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) { acc[0][mt] = zero4; acc[1][mt] = zero4; }
#pragma unroll
      for (int q = 0; q < M::QH; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) {
          const float fr = F.b3(mt, q);               // one fragment read, two MFMAs
          acc[0][mt] = mfma4(fr, d3[0][q], acc[0][mt]);
          acc[1][mt] = mfma4(fr, d3[1][q], acc[1][mt]);
        }
      float d2[2][M::QW];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < M::QW; ++q)
          d2[t][q] = acc[t][q / 4][q % 4] * dact_stored<C::ACT, DROP>(a2[t][q]);
      wave_lds_sync();

      // ---- layer 2
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        img_write<M::QW>(img_d[t], d2[t], g, c);
        img_write<M::Q1>(img_a[t], a1[t], g, c);
      }
      wave_lds_sync();
      dw_accumulate<M::MT1, NT1>(img_d[0], img_a[0], G2, g, c);
      dw_accumulate<M::MT1, NT1>(img_d[1], img_a[1], G2, g, c);
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) { acc[0][mt] = zero4; acc[1][mt] = zero4; }
#pragma unroll
      for (int q = 0; q < M::QW; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) {
          const float fr = F.b2(mt, q);
          acc[0][mt] = mfma4(fr, d2[0][q], acc[0][mt]);
          acc[1][mt] = mfma4(fr, d2[1][q], acc[1][mt]);
        }
      float d1[2][M::QW];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < M::QW; ++q)
          d1[t][q] = acc[t][q / 4][q % 4] * dact_stored<C::ACT, DROP>(a1[t][q]);
      wave_lds_sync();

      // ---- layer 1
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        img_write<M::QW>(img_d[t], d1[t], g, c);
        img_write<M::Q0>(img_a[t], b0[t], g, c);
      }
      wave_lds_sync();
      dw_accumulate<M::MT1, NT0>(img_d[0], img_a[0], G1, g, c);
      dw_accumulate<M::MT1, NT0>(img_d[1], img_a[1], G1, g, c);
      f32x4 acch[2][M::MTH];
#pragma unroll
      for (int mt = 0; mt < M::MTH; ++mt) {
        f32x4 e0 = zero4, e1 = zero4;
#pragma unroll
        for (int q = 0; q < M::QW; ++q) {
          const float fr = F.b1(mt, q);
          e0 = mfma4(fr, d1[0][q], e0);
          e1 = mfma4(fr, d1[1][q], e1);
        }
        acch[0][mt] = e0;
        acch[1][mt] = e1;
      }
      // adjoint of the state: lam += (W1^T delta1)[h rows] * (1 - tanh(h)^2)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < M::QH; ++q) {
          const float th = b0[t][q];  // = tanh(h) wherever unit 4q + g < H
          const float dth = (4 * q + g) < C::H ? 1.0f - th * th : 0.0f;
          lam[t][q] = fmaf(acch[t][q / 4][q % 4], dth, lam[t][q]);
        }
      wave_lds_sync();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float* out = valid[t] ? a.lam_start + (size_t)it[t].r * C::H : trash;
#pragma unroll
      for (int q = 0; q < M::QH; ++q) {
        const int u = 4 * q + g;
        float* dst = u < C::H ? out + u : trash;
        *dst = lam[t][q];
      }
    }
  }

  // ---- flush (as ode3_bwd_single): one slab row per block
  constexpr int NG = M::MTH * NT1 + M::MT1 * NT1 + M::MT1 * NT0;
  static_assert(3 * NG * 64 * 4 <= OdeBwdPairLds<C>::FLOATS, "tile reduction does not fit the LDS");
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
  const float ik = DROP ? a.dc.inv_keep : 1.0f;
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
          if (ui < M::W) W2[uo * M::W + ui] = ik * G2[mt][nt][r];
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
          if (ui < M::W) W3[uo * M::W + ui] = ik * G3[mt][nt][r];
          else if (ui == M::W) b3[uo] = G3[mt][nt][r];
        }
      }
    }
}


}  // namespace njode
