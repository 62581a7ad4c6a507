// njode_mfma_split.h -- ODE-evolve kernels of the segment plan with one tile of 16 chains
// spread over the FOUR waves (= four SIMDs) of a 256-thread block.
//
// Why: a segment is a serial chain of Euler steps, and in njode_mfma.h one wave issues every
// MFMA of a step (81 forward / 241 backward) from one SIMD, ~3 us / ~8 us per step.  The
// longest segments of a batch (60-90 steps) therefore set the kernel time however many waves
// run beside them (profiles/r01_v6: a batch of equal-length segments is 1.5x faster than the
// real length distribution).  Here wave w owns output tile w (hidden units 16w .. 16w+15) of
// the two hidden layers and every product whose result lives on those units:
//
//   forward   L1 (own tile, 4 MFMA) -> all-gather a1 through LDS -> L2 (own tile, 13) ->
//             L3 split over K (own units' k-steps, <= 4) -> reduce the 4 partial tiles
//   backward  + W3^T d3 (own tile, 3) -> all-gather d2 -> W2^T d2 (own tile, 13) ->
//             W1^T d1 split over K (<= 4) -> reduce;  dW: G3 column tile w (4), G2 row
//             tile w (16), G1 row tile w (4)
//
// = 21 / 61 MFMAs per wave and step instead of 81 / 241, the activations' tanh and dropout
// only for the own 4 registers, weights as 37 register-resident A-fragments per wave (no
// fragment traffic at all), 2 / 3 s_barriers per step.  The exchange buffers double as the
// [unit][chain] images the dW products read (njode_mfma.h), so the all-gather costs no
// extra LDS traffic.  Every wave keeps its own copy of the state h and the adjoint.
//
// A four-wave block runs a tile-step ~2.3x faster than one wave does, but at ~20 % lower
// throughput per SIMD (barriers, LDS round trips), so it does not replace the one-wave form:
// the MIXED kernels at the end of this file give the longest tiles to a few four-wave blocks
// and the bulk to one-wave workers, with the split point chosen on the device so that both
// groups finish together (k_split_point, njode_api.hip; profiles/r01_mixed_sweep.txt).
//
// Only for shapes with four hidden tiles and one state tile (32 < W <= 63, H <= 16): the demo
// networks.  Dropout masks are those of the one-wave kernels, whatever role a tile runs in.
#pragma once
#include "njode_mfma.h"
#include "njode_ode2.h"
#include "njode_plan.h"

namespace njode {

template <class C, bool TWO = (C::NH == 2)> struct SplitOk { static constexpr bool value = false; };
template <class C> struct SplitOk<C, true> {   // (MF<C> only exists for two hidden layers)
  static constexpr bool value = MF<C>::MT1 == 4 && MF<C>::MTH == 1 && !C::MASKED && !C::RNN;
};

constexpr int XROWS = 64;                     // shared exchange images: one row per hidden unit
constexpr int XFLOATS = XROWS * IMG_STRIDE;
constexpr int TROWS = 16;                     // wave-private images: one tile
constexpr int TFLOATS = TROWS * IMG_STRIDE;

// this wave's A-fragments (k_pack_frags table of njode_mfma.h)
template <class C, bool BWD> struct SplitFrags {
  using M = MF<C>;
  float A1[M::Q0], A2[M::Q1], A3[4];
  float B3[BWD ? M::QH : 1], B2[BWD ? M::QW : 1], B1[BWD ? 4 : 1];
  NJ_DEV void load(const float* frag, int w, int lane) {
#pragma unroll
    for (int q = 0; q < M::Q0; ++q) A1[q] = frag[(M::F1 + w * M::Q0 + q) * 64 + lane];
#pragma unroll
    for (int q = 0; q < M::Q1; ++q) A2[q] = frag[(M::F2 + w * M::Q1 + q) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = 4 * w + r;
      A3[r] = q < M::Q1 ? frag[(M::F3 + (q < M::Q1 ? q : 0)) * 64 + lane] : 0.0f;
    }
    if constexpr (BWD) {
#pragma unroll
      for (int q = 0; q < M::QH; ++q) B3[q] = frag[(M::B3 + w * M::QH + q) * 64 + lane];
#pragma unroll
      for (int q = 0; q < M::QW; ++q) B2[q] = frag[(M::B2 + w * M::QW + q) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 4 * w + r;
        B1[r] = q < M::QW ? frag[(M::B1 + (q < M::QW ? q : 0)) * 64 + lane] : 0.0f;
      }
    }
  }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (its
// fence covers global memory), which would put the latency of the next step's prefetched
// checkpoint / schedule loads on the critical path of every Euler step.
NJ_DEV void block_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Keep bits of this wave's 2 x 4 hidden registers -- the SAME masks the one-wave kernels
// draw (njode_mfma.h: unit 4q + g is bit q of lane group g's stream, NWD words per layer), so a
// tile may run four-wide in one kernel and one wave wide in another.  Wave w owns bits
// 4w .. 4w+3 = words 2w, 2w+1 of each layer's words.
template <class C, bool DROP>
NJ_DEV void split_keep_bits(const KArgs& a, int b, int k, int g, int w, uint32_t& k1, uint32_t& k2) {
  constexpr int NWD = (MF<C>::Q1 + 1) / 2;   // words per layer in the one-wave kernels
  k1 = k2 = 0;
  if constexpr (DROP) {
    const unsigned long long gid = a.gid0 + b;
    uint32_t st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1),
                             (uint32_t)k, NET_ODE);
    auto skip = [&](int n) {
      for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 17; st ^= st << 5; }
    };
    skip(2 * w);
    k1 = keep_bits<4>(st, a.dc.thr16);
    skip(NWD - 2);                         // rest of layer 1's words, words 0 .. 2w-1 of layer 2
    k2 = keep_bits<4>(st, a.dc.thr16);
  }
}

// The keep bits of the four-wave forward AHEAD of the kernel.  In the four-wave role a Euler
// step is a chain of two exchanges on four SIMDs that each run ONE wave: every instruction of the
// step is on its critical path, and split_keep_bits -- three hash rounds, up to eleven serial
// xorshift words to reach the wave's own four, the bit assembly: ~130 VALU instructions -- was a
// fifth of the step although it depends on nothing the chain computes.  drop_bits_tile_steps
// draws, for the tile-steps of the T longest tiles, the bits of ALL four waves at once (the whole
// lane-group stream of the one-wave kernels: k1 | k2 << 16, bit q = register q = unit 4q + g),
// one dword per lane; it runs in spare blocks of the fragment-pack launch, i.e. in parallel over
// the chip and off every chain.  The forward then loads one dword per step (one step ahead).
// Same masks, bit for bit, as split_keep_bits / the one-wave kernels.
template <class C>
NJ_DEV void drop_bits_tile_steps(const KArgs& a, int worker, int n_workers) {
  using M = MF<C>;
  constexpr int CH = 8;                                  // Euler steps per work item
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int T = (int)a.base_s[a.K + 2];                  // split point of the forward
  const int n_ch = (a.K + CH - 1) / CH;
  const long long n_work = (long long)T * n_ch;
  for (long long i = worker; i < n_work; i += n_workers) {
    const int tile = (int)(i / n_ch), s0 = (int)(i % n_ch) * CH;
    const int j = tile * 16 + c;
    const bool valid = j < a.n_obs;
    const int r = a.order[valid ? j : 0];
    const int n = valid ? a.item_len[r] : 0;
    const int nmax = uniform(wave_max(n));
    if (s0 >= nmax) continue;
    const int b = a.obs_idx[r], kbeg = a.item_kbeg[r];
    const unsigned long long gid = a.gid0 + b;
    for (int s = s0; s < s0 + CH && s < nmax; ++s) {
      const int k = s < n ? kbeg + s : 0;
      uint32_t st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1),
                               (uint32_t)k, NET_ODE);
      const uint32_t k1 = keep_bits<M::Q1>(st, a.dc.thr16);
      const uint32_t k2 = keep_bits<M::Q1>(st, a.dc.thr16);
      a.dbits[(size_t)(sload_ll(a.base16_s, s) / 16 + tile) * 64 + lane] = k1 | (k2 << 16);
    }
  }
}

// activation of the own tile: al[r] = act(acc[r]) (+ dropout); the bias unit (unit W) is 1
// (au: the form the forward stores for the backward -- no 1 / (1 - p), a dropped unit as -0.0f)
template <class C, bool DROP>
NJ_DEV void split_hidden(const f32x4& acc, float (&al)[4], float (&au)[4], uint32_t keep, float inv_keep,
                         int g, int w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = act_f<C::ACT>(acc[r]), u = v;
    if constexpr (DROP) {
      const bool k = (keep >> r) & 1;
      u = k ? v : -0.0f;
      v = k ? v * inv_keep : 0.0f;
    }
    al[r] = v;
    au[r] = u;
  }
  constexpr int QB = C::W / 4, GB = C::W % 4;
  if (w == QB / 4) {
    al[QB % 4] = g == GB ? 1.0f : al[QB % 4];
    au[QB % 4] = g == GB ? 1.0f : au[QB % 4];
  }
}
template <class C, bool DROP>
NJ_DEV void split_delta(const f32x4& acc, const float (&al)[4], float (&dl)[4], uint32_t keep,
                        float inv_keep, float keepf, int g, int w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float gsum = acc[r];
    float d;
    if constexpr (DROP) d = ((keep >> r) & 1) ? gsum * inv_keep * dact_f<C::ACT>(al[r] * keepf) : 0.0f;
    else d = gsum * dact_f<C::ACT>(al[r]);
    dl[r] = (16 * w + 4 * r + g) < C::W ? d : 0.0f;   // bias / padding units carry no delta
  }
}

// own 4 registers -> rows 16w + 4r + g of a shared [unit][chain] image
NJ_DEV void split_put(lfp X, const float (&v)[4], int g, int c, int w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) X[(16 * w + 4 * r + g) * IMG_STRIDE + c] = v[r];
}
template <int NQ> NJ_DEV void split_get(lfp X, float (&v)[NQ], int g, int c) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) v[q] = X[(4 * q + g) * IMG_STRIDE + c];
}

// wave w's share of a tile's step record (StepRec, njode_ode2.h): its registers 4w .. 4w+3 of
// a1 / a2 are quad w of group B / A; the wave that holds the layers' leftover registers (wave
// NF; wave 3 when there are none) also stores group A's tail list -- those leftovers and the
// state h, of which every wave keeps a copy -- and a1's leftovers.  No branch around the stores:
// what a wave does not own goes to the `trash` block, so that the compiler can count them
// (see ode2_fwd_single).
template <class C>
NJ_DEV void split_rec_store(float* blk, float* trash, int lane, int w, const float (&a1u)[4],
                            const float (&a2u)[4], const float (&h)[MF<C>::QH]) {
  using R = StepRec<C>;
  float* bq = w < R::NF ? blk : trash;
  quad_store(bq + w * 256 + lane * 4, a2u[0], a2u[1], a2u[2], a2u[3]);
  quad_store(bq + R::GA + w * 256 + lane * 4, a1u[0], a1u[1], a1u[2], a1u[3]);
  float a2x[MF<C>::Q1];
#pragma unroll
  for (int q = 0; q < MF<C>::Q1; ++q) a2x[q] = 0.0f;
#pragma unroll
  for (int i = 0; i < R::L1; ++i) a2x[4 * R::NF + i] = a2u[i];
  rec_tail_store<C>(w == R::WT ? blk : trash, lane, a2x, h);
  if constexpr (R::L1 > 0) {
    float* bs = w == R::NF ? blk : trash;
#pragma unroll
    for (int i = 0; i < R::L1; ++i) bs[R::B1S + i * 64 + lane] = a1u[i];
  }
}
// the loaded words of a step record, RAW: they cross the loop's back edge as they are and are
// sorted into (a1u, a2u, h) where the step starts (a select next to a load would put the
// load's s_waitcnt there)
template <class C> struct SplitRecRaw {
  using R = StepRec<C>;
  f32x4 qa, qb;
  float a2x[MF<C>::Q1], h[MF<C>::QH], s1[R::L1 > 0 ? R::L1 : 1];
  NJ_DEV void zero() {
    qa = qb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < MF<C>::Q1; ++q) a2x[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < MF<C>::QH; ++q) h[q] = 0.0f;
    s1[0] = 0.0f;
  }
  NJ_DEV void load(const float* blk, int lane, int w) {
    const int wq = w < R::NF ? w : 0;
    qa = *(const f32x4*)(blk + wq * 256 + lane * 4);
    qb = *(const f32x4*)(blk + R::GA + wq * 256 + lane * 4);
    rec_tail_load<C>(blk, lane, a2x, h);
#pragma unroll
    for (int i = 0; i < R::L1; ++i) s1[i] = blk[R::B1S + i * 64 + lane];
  }
  NJ_DEV void unpack(int w, float (&a1u)[4], float (&a2u)[4], float (&hh)[MF<C>::QH]) const {
    const bool own = w < R::NF;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float l2 = r < R::L1 ? a2x[4 * R::NF + (r < R::L1 ? r : 0)] : 0.0f;
      const float l1 = r < R::L1 ? s1[r < R::L1 ? r : 0] : 0.0f;
      a2u[r] = own ? qa[r] : l2;
      a1u[r] = own ? qb[r] : l1;
    }
#pragma unroll
    for (int q = 0; q < MF<C>::QH; ++q) hh[q] = h[q];
  }
};

// forward of the two hidden layers for the own tile; leaves a1 (all units) gathered
template <class C, bool DROP, bool BWD>
NJ_DEV void split_hidden_layers(const SplitFrags<C, BWD>& F, lfp X1, const float (&b0)[MF<C>::Q0],
                                float (&a1l)[4], float (&a2l)[4], float (&a1u)[4], float (&a2u)[4],
                                uint32_t k1, uint32_t k2, float inv_keep, int g, int c, int w) {
  using M = MF<C>;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc = z;
#pragma unroll
  for (int q = 0; q < M::Q0; ++q) acc = mfma4(F.A1[q], b0[q], acc);
  split_hidden<C, DROP>(acc, a1l, a1u, k1, inv_keep, g, w);
  split_put(X1, a1l, g, c, w);
  block_lds_barrier();                               // (a) a1 of all four tiles is in X1
  float a1[M::Q1];
  split_get<M::Q1>(X1, a1, g, c);
  // the gather is issued as one batch before the products: left alone the compiler emits
  // ds_read2 -> s_waitcnt lgkmcnt(0) -> two MFMAs, seven LDS round trips in a row
  __builtin_amdgcn_sched_barrier(0);
  // two accumulators halve the dependent chain of the 13 k-steps
  f32x4 acc0 = z, acc1 = z;
#pragma unroll
  for (int q = 0; q < M::Q1; q += 2) {
    acc0 = mfma4(F.A2[q], a1[q], acc0);
    if (q + 1 < M::Q1) acc1 = mfma4(F.A2[q + 1], a1[q + 1], acc1);
  }
  acc = acc0 + acc1;
  split_hidden<C, DROP>(acc, a2l, a2u, k2, inv_keep, g, w);
}

// partial product over the own k-steps, summed over the four waves in fixed order
template <int NQ>
NJ_DEV void split_reduce(lfp PR, const f32x4& part, float (&out)[NQ], int lane, int w) {
#pragma unroll
  for (int r = 0; r < NQ; ++r) PR[(w * NQ + r) * 64 + lane] = part[r];
  block_lds_barrier();
#pragma unroll
  for (int r = 0; r < NQ; ++r) {
    float s = PR[(0 * NQ + r) * 64 + lane];
    s += PR[(1 * NQ + r) * 64 + lane];
    s += PR[(2 * NQ + r) * 64 + lane];
    s += PR[(3 * NQ + r) * 64 + lane];
    out[r] = s;
  }
}

template <class C> struct OdeFwdSplitLds { static constexpr int FLOATS = XFLOATS + 4 * MF<C>::QH * 64; };
// B (split): Euler evolve; block `worker` of `n_workers` walks the tiles [tile0, tile1)
template <class C, bool DROP, bool TAIL, bool SAVE>
NJ_DEV void ode_fwd_split(const KArgs& a, lfp lds_raw, int worker, int n_workers, int tile0, int tile1) {
  static_assert(!(TAIL && SAVE), "tail items are never checkpointed");
  using M = MF<C>;
  lfp X1 = lds_raw, PR = X1 + XFLOATS;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < XFLOATS; i += 256) lds_raw[i] = 0.0f;
  SplitFrags<C, false> F;
  F.load(a.frag, w, lane);
  In0Const<C> K0;
  K0.init(g);
  __syncthreads();
  const bool bits_ahead = DROP && !TAIL && a.dbits_ready != 0;   // (wave-uniform)

  const int n_items = TAIL ? a.B : a.n_obs;
  const int n_tiles = tile1 - tile0;
  float* const trash = a.trash + lane * C::H;
  for (int round = 0; round * n_workers < n_tiles; ++round) {
    const int rel = snake_tile(round, worker, n_workers);
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
    const int nmax = uniform(wave_max(it.n));   // (scalar step loop: see ode2_fwd_single)
    float dt_r = 0.0f, t_r = 0.0f;
    uint32_t kb_r = 0;
    long long b16_n = 0;
    if (nmax > 0) {
      const int k0 = it.n > 0 ? it.kbeg : 0;
      dt_r = a.step_dt[k0];
      t_r = a.step_t[k0];
      b16_n = (SAVE || bits_ahead) ? sload_ll(a.base16_s, 0) : 0;
      if (bits_ahead) kb_r = a.dbits[(size_t)(b16_n / 16 + tile) * 64 + lane];
    }
    if constexpr (SAVE) {
      if (a.item_pack) {   // (for the backward's tile prologue: Item::store_pack; whichever role runs the tile there)
        if (w == 0 && g == 0) it.store_pack(a.item_pack, j);
        const long long bl = sload_ll(a.base16_s, nmax > 0 ? nmax - 1 : 0);
        if (threadIdx.x == 0) a.tile_last[tile] = bl;
      }
    }
    vm_drain();
    for (int s = 0; s < nmax; ++s) {
      const bool active = s < it.n;
      const int k = active ? it.kbeg + s : 0;
      const float dt = active ? dt_r : 0.0f, t = t_r;
      const uint32_t kb = kb_r;
      const long long b16 = b16_n;
      {
        const int sn = s + 1 < nmax ? s + 1 : s;
        const int kn = sn < it.n ? it.kbeg + sn : 0;
        dt_r = a.step_dt[kn];
        t_r = a.step_t[kn];
        if (SAVE || bits_ahead) b16_n = sload_ll(a.base16_s, sn);
        if (bits_ahead) kb_r = a.dbits[(size_t)(b16_n / 16 + tile) * 64 + lane];
      }
      float b0[M::Q0], a1l[4], a2l[4], a1u[4], a2u[4];
      in0_fill_c<C, 0>(b0, h, it.tx, it.tau, t - it.tau, g, K0);
      uint32_t k1 = 0, k2 = 0;
      if constexpr (DROP) {
        if (bits_ahead) {                     // (uniform) drawn ahead: drop_bits_tile_steps
          k1 = (kb >> (4 * w)) & 15u;
          k2 = (kb >> (16 + 4 * w)) & 15u;
        } else {
          split_keep_bits<C, DROP>(a, it.b, k, g, w, k1, k2);
        }
      }
      split_hidden_layers<C, DROP, false>(F, X1, b0, a1l, a2l, a1u, a2u, k1, k2, a.dc.inv_keep, g, c, w);
      if constexpr (SAVE) split_rec_store<C>(rec_block<C>(a.act, b16, tile), a.trash, lane, w, a1u, a2u, h);
      f32x4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * w + r < M::Q1) part = mfma4(F.A3[r], a2l[r], part);
      float f[M::QH];
      split_reduce<M::QH>(PR, part, f, lane, w);      // (b)
#pragma unroll
      for (int q = 0; q < M::QH; ++q) h[q] = fmaf(dt, f[q], h[q]);   // dt = 0: inactive
    }
    if (w == 0) {
      float* out = valid ? (TAIL ? a.hT + (size_t)it.b * C::H : a.h_end + (size_t)it.r * C::H) : trash;
#pragma unroll
      for (int q = 0; q < M::QH; ++q) {
        const int u = 4 * q + g;
        float* dst = u < C::H ? out + u : trash;
        *dst = h[q];
      }
    }
  }
}

// shared: X1 (a1 / its image), X2 (d2 / its image), reduce buffer; per wave: images of d3,
// own a2 tile, own d1 tile, b0
template <class C> struct OdeBwdSplitLds {
  static constexpr int FLOATS = 2 * XFLOATS + 4 * MF<C>::QH * 64 + 4 * 4 * TFLOATS + 64;   // (+ the tile queue's hand-over word)
};
// C (split): reverse Euler sweep, d loss / d ODE params.  Block `worker` of `n_workers` walks
// the tiles [tile0, tile1); one slab row per block, wave w flushes the tiles it owns.
template <class C, bool DROP>
NJ_DEV void ode_bwd_split(const KArgs& a, lfp lds_raw, int worker, int n_workers, int tile0, int tile1,
                          int slab_row) {
  using M = MF<C>;
  using NL = typename C::Ode;
  constexpr int NT1 = 4;
  static_assert((M::W + 1 + 15) / 16 == NT1 && (M::IN0 + 1 + 15) / 16 == 1,
                "split kernels: 4 column tiles of [a, 1], one of [in0, 1]");
  lfp X1 = lds_raw, X2 = X1 + XFLOATS, PR = X2 + XFLOATS;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lfp imgD3 = PR + 4 * M::QH * 64 + w * 4 * TFLOATS, imgA2 = imgD3 + TFLOATS, imgD1 = imgA2 + TFLOATS,
      imgB0 = imgD1 + TFLOATS;
  for (int i = threadIdx.x; i < 2 * XFLOATS + 4 * M::QH * 64 + 4 * 4 * TFLOATS; i += 256) lds_raw[i] = 0.0f;
  SplitFrags<C, true> F;
  F.load(a.frag, w, lane);
  __syncthreads();

  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 G3[1][1] = {{z}}, G2[1][NT1] = {{z, z, z, z}}, G1[1][1] = {{z}};
  float* const trash = a.trash + lane * C::H;
  const int n_tiles = tile1 - tile0;
  for (int round = 0; round * n_workers < n_tiles; ++round) {
    const int rel = snake_tile(round, worker, n_workers);
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
      float b0[M::Q0], a1l[4], a2l[4];
      in0_fill<C, 0>(b0, h, it.tx, it.tau, t - it.tau, g);
      uint32_t k1, k2;
      split_keep_bits<C, DROP>(a, it.b, k, g, w, k1, k2);
      float a1u_[4], a2u_[4];
      split_hidden_layers<C, DROP, true>(F, X1, b0, a1l, a2l, a1u_, a2u_, k1, k2, a.dc.inv_keep, g, c, w);  // (a)

      // ---- layer 3: delta3 = dt * lam; dW3 column tile w; W3^T delta3 for the own units
      float d3[M::QH];
#pragma unroll
      for (int q = 0; q < M::QH; ++q) d3[q] = dt * lam[q];
      img_write<M::QH>(imgD3, d3, g, c);
#pragma unroll
      for (int r = 0; r < 4; ++r) imgA2[(4 * r + g) * IMG_STRIDE + c] = a2l[r];
      img_write<M::Q0>(imgB0, b0, g, c);
      f32x4 acc = z;
#pragma unroll
      for (int q = 0; q < M::QH; ++q) acc = mfma4(F.B3[q], d3[q], acc);
      float d2l[4], d1l[4];
      split_delta<C, DROP>(acc, a2l, d2l, k2, a.dc.inv_keep, a.keep, g, w);
      split_put(X2, d2l, g, c, w);
      wave_lds_sync();
      dw_accumulate<1, 1>(imgD3, imgA2, G3, g, c);
      block_lds_barrier();                             // (b) d2 of all four tiles is in X2

      // ---- layer 2: W2^T delta2 for the own units; dW2 row tile w
      float d2[M::QW];
      split_get<M::QW>(X2, d2, g, c);
      __builtin_amdgcn_sched_barrier(0);   // (as in split_hidden_layers)
      f32x4 acc0 = z, acc1 = z;
#pragma unroll
      for (int q = 0; q < M::QW; q += 2) {
        acc0 = mfma4(F.B2[q], d2[q], acc0);
        if (q + 1 < M::QW) acc1 = mfma4(F.B2[q + 1], d2[q + 1], acc1);
      }
      dw_accumulate<1, NT1>(X2 + 16 * w * IMG_STRIDE, X1, G2, g, c);
      acc = acc0 + acc1;
      split_delta<C, DROP>(acc, a1l, d1l, k1, a.dc.inv_keep, a.keep, g, w);

      // ---- layer 1: dW1 row tile w; W1^T delta1 split over the own k-steps
#pragma unroll
      for (int r = 0; r < 4; ++r) imgD1[(4 * r + g) * IMG_STRIDE + c] = d1l[r];
      f32x4 part = z;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * w + r < M::QW) part = mfma4(F.B1[r], d1l[r], part);
      wave_lds_sync();
      dw_accumulate<1, 1>(imgD1, imgB0, G1, g, c);
      float gh[M::QH];
      split_reduce<M::QH>(PR, part, gh, lane, w);      // (c)
#pragma unroll
      for (int q = 0; q < M::QH; ++q) {
        const float th = b0[q];  // = tanh(h) wherever unit 4q + g < H
        const float dth = (4 * q + g) < C::H ? 1.0f - th * th : 0.0f;
        lam[q] = fmaf(gh[q], dth, lam[q]);
      }
    }
    if (w == 0) {
      float* out = valid ? a.lam_start + (size_t)it.r * C::H : trash;
#pragma unroll
      for (int q = 0; q < M::QH; ++q) {
        const int u = 4 * q + g;
        float* dst = u < C::H ? out + u : trash;
        *dst = lam[q];
      }
    }
  }

  // ---- flush the tiles this wave owns into the block's slab row (parameter layout)
  float* slab = a.slab + (size_t)slab_row * C::P + C::OFF_ODE;
  float *W1 = slab + NL::woff(0), *b1 = slab + NL::boff(0), *W2 = slab + NL::woff(1),
        *b2 = slab + NL::boff(1), *W3 = slab + NL::woff(2), *b3 = slab + NL::boff(2);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int uo = 16 * w + 4 * g + r;
    if (uo < M::W) {
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) {
        const int ui = 16 * nt + c;
        if (ui < M::W) W2[uo * M::W + ui] = G2[0][nt][r];
        else if (ui == M::W) b2[uo] = G2[0][nt][r];
      }
      if (c < M::IN0) W1[uo * M::IN0 + M::col0(c)] = G1[0][0][r];
      else if (c == M::IN0) b1[uo] = G1[0][0][r];
    }
    const int uh = 4 * g + r, ui = 16 * w + c;
    if (uh < C::H) {
      if (ui < M::W) W3[uh * M::W + ui] = G3[0][0][r];
      else if (ui == M::W) b3[uh] = G3[0][0][r];
    }
  }
}


// delta of the own units from the stored activation (unscaled, dropped = -0.0f)
template <class C, bool DROP>
NJ_DEV void split_delta_stored(const f32x4& acc, const float (&au)[4], float (&dl)[4], float ik, int g, int w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float d = acc[r] * ik * dact_stored<C::ACT, DROP>(au[r]);
    dl[r] = (16 * w + 4 * r + g) < C::W ? d : 0.0f;   // bias / padding units carry no delta
  }
}
// this wave's transposed-product fragments only (the backward on stored activations)
template <class C> struct SplitFragsT {
  using M = MF<C>;
  float B3[M::QH], B2[M::QW], B1[4];
  NJ_DEV void load(const float* frag, int w, int lane) {
#pragma unroll
    for (int q = 0; q < M::QH; ++q) B3[q] = frag[(M::B3 + w * M::QH + q) * 64 + lane];
#pragma unroll
    for (int q = 0; q < M::QW; ++q) B2[q] = frag[(M::B2 + w * M::QW + q) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = 4 * w + r;
      B1[r] = q < M::QW ? frag[(M::B1 + (q < M::QW ? q : 0)) * 64 + lane] : 0.0f;
    }
  }
};

// C (split, stored activations): as ode_bwd_split, the hidden activations loaded
// QUEUE (njode_ode2.h, tile queue): the block pops tiles [tile0, tile1) from tile_q[0], then -- a
// four-wave block runs any tile -- helps with the bulk's [tile1, n_all) from tile_q[1]
template <class C, bool DROP, bool QUEUE = false>
NJ_DEV void ode3_bwd_split(const KArgs& a, lfp lds_raw, int worker, int n_workers, int tile0, int tile1,
                          int slab_row, int n_all = 0) {
  using M = MF<C>;
  using NL = typename C::Ode;
  constexpr int NT1 = 4;
  static_assert((M::W + 1 + 15) / 16 == NT1 && (M::IN0 + 1 + 15) / 16 == 1,
                "split kernels: 4 column tiles of [a, 1], one of [in0, 1]");
  lfp X1 = lds_raw, X2 = X1 + XFLOATS, PR = X2 + XFLOATS;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lfp imgD3 = PR + 4 * M::QH * 64 + w * 4 * TFLOATS, imgA2 = imgD3 + TFLOATS, imgD1 = imgA2 + TFLOATS,
      imgB0 = imgD1 + TFLOATS;
  for (int i = threadIdx.x; i < 2 * XFLOATS + 4 * M::QH * 64 + 4 * 4 * TFLOATS; i += 256) lds_raw[i] = 0.0f;
  SplitFragsT<C> F;
  F.load(a.frag, w, lane);
  In0Const<C> K0;
  K0.init(g);
  __syncthreads();

  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 G3[1][1] = {{z}}, G2[1][NT1] = {{z, z, z, z}}, G1[1][1] = {{z}};
  float* const trash = a.trash + lane * C::H;
  const int n_tiles = tile1 - tile0;
  BWD_STAMP(1, wall_clock64());
  // (QUEUE: wave 0 pops -- one tile ahead, so the atomic's round trip hides behind the sweep -- and
  // hands the tile to the block through one LDS word)
  int __attribute__((address_space(3)))* qword = (int __attribute__((address_space(3)))*)(lds_raw + OdeBwdSplitLds<C>::FLOATS - 64);
  // (QUEUE) wave 0 pops: first tile = the block's own index (static, no burst at launch), then
  // tile_q[0] counts on from n_workers through [tile0, tile1), then tile_q[1] from the bulk's
  // worker count through [tile1, n_all); the pop for the NEXT tile is issued, unwaited, when a
  // tile's prologue is done and read one tile later (njode_ode2.h, queue_pop_issue)
  const int n_bulk_workers = QUEUE ? ((int)gridDim.x - n_workers) * 4 : 0;
  int q_phase = 0, q_raw = 0;
  auto resolve = [&](int raw) -> int {          // tile of a returned pop; may need a second pop
    if (q_phase == 0) {
      const int t = n_workers + queue_value(raw);
      if (t < n_tiles) return tile0 + t;
      q_phase = 1;
      return tile1 + n_bulk_workers + queue_value(queue_pop_issue(a.tile_q + 1));
    }
    return tile1 + n_bulk_workers + queue_value(raw);
  };
  int q_next = 0;
  if constexpr (QUEUE) {
    if (worker < n_tiles) q_next = tile0 + worker;
    else { q_phase = 1; if (w == 0) q_next = tile1 + n_bulk_workers + queue_value(queue_pop_issue(a.tile_q + 1)); }
  }
  int n_done = 0, n_steps_done = 0;
  for (int round = 0; QUEUE || round * n_workers < n_tiles; ++round) {
    int tile;
    if constexpr (QUEUE) {
      if (round > 0 && w == 0) q_next = resolve(q_raw);
      if (w == 0 && lane == 0) *qword = q_next;
      block_lds_barrier();
      tile = uniform(*qword);
      block_lds_barrier();
      if (tile >= n_all) break;
    } else {
      const int rel = snake_tile(round, worker, n_workers);
      if (rel >= n_tiles) continue;
      tile = tile0 + rel;
    }
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
    const int nmax = uniform(wave_max(it.n));   // (scalar step loop, scalar record address)
    ++n_done;
    n_steps_done += nmax;
    // the record and the schedule values of step s - 1 are loaded while step s runs, RAW
    SplitRecRaw<C> nx;
    nx.zero();
    float dt_r = 0.0f, t_r = 0.0f;
    auto fetch = [&](int s) {
      const int kk = s < it.n ? it.kbeg + s : 0;
      nx.load(rec_block<C>(a.act, sload_ll(a.base16_s, s), tile), lane, w);
      dt_r = a.step_dt[kk];
      t_r = a.step_t[kk];
    };
    if (nmax > 0) fetch(nmax - 1);
    vm_drain();
    if constexpr (QUEUE) { if (w == 0) q_raw = queue_pop_issue(a.tile_q + (q_phase == 0 ? 0 : 1)); }
    for (int s = nmax - 1; s >= 0; --s) {
      float h[M::QH], a1u[4], a2u[4];
      nx.unpack(w, a1u, a2u, h);
      const float dt = s < it.n ? dt_r : 0.0f, t = t_r;
      fetch(s > 0 ? s - 1 : 0);
      float b0[M::Q0], a1l[4], a2l[4];
      in0_fill_c<C, 0>(b0, h, it.tx, it.tau, t - it.tau, g, K0);
      // the forward's activations are loaded, not recomputed: with the factor 1 / (1 - p) they
      // are the images' operands; all-gather a1 for the dW2 product
      const float ik = DROP ? a.dc.inv_keep : 1.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { a1l[r] = a1u[r] * ik; a2l[r] = a2u[r] * ik; }
      constexpr int QBb = C::W / 4, GBb = C::W % 4;
      if (w == QBb / 4) {    // the bias unit is 1 (not 1 / (1 - p))
        a1l[QBb % 4] = g == GBb ? 1.0f : a1l[QBb % 4];
        a2l[QBb % 4] = g == GBb ? 1.0f : a2l[QBb % 4];
      }
      // (round 4: no barrier here.  X1 -- a1 of all four tiles -- is only read by the dW2
      // product behind barrier (b), and it is only rewritten behind barrier (c), by when every
      // wave has finished that product: the all-gather of a1 rides on the barriers the delta
      // chain needs anyway)
      split_put(X1, a1l, g, c, w);

      // ---- layer 3: delta3 = dt * lam; dW3 column tile w; W3^T delta3 for the own units
      float d3[M::QH];
#pragma unroll
      for (int q = 0; q < M::QH; ++q) d3[q] = dt * lam[q];
      f32x4 acc = z;
#pragma unroll
      for (int q = 0; q < M::QH; ++q) acc = mfma4(F.B3[q], d3[q], acc);
      img_write<M::QH>(imgD3, d3, g, c);
#pragma unroll
      for (int r = 0; r < 4; ++r) imgA2[(4 * r + g) * IMG_STRIDE + c] = a2l[r];
      img_write<M::Q0>(imgB0, b0, g, c);
      float d2l[4], d1l[4];
      split_delta_stored<C, DROP>(acc, a2u, d2l, ik, g, w);
      split_put(X2, d2l, g, c, w);
      block_lds_barrier();                             // (b) d2 (and a1) of all four tiles are in X2 (X1)

      // ---- layer 2: W2^T delta2 for the own units; dW3 column tile w, dW2 row tile w
      float d2[M::QW];
      split_get<M::QW>(X2, d2, g, c);
      __builtin_amdgcn_sched_barrier(0);   // (as in split_hidden_layers)
      // (the dW3 product -- its images are this wave's own, written before the barrier -- runs
      // while the gather of d2 is in flight, not in front of the barrier)
      dw_accumulate<1, 1>(imgD3, imgA2, G3, g, c);
      f32x4 acc0 = z, acc1 = z;
#pragma unroll
      for (int q = 0; q < M::QW; q += 2) {
        acc0 = mfma4(F.B2[q], d2[q], acc0);
        if (q + 1 < M::QW) acc1 = mfma4(F.B2[q + 1], d2[q + 1], acc1);
      }
      dw_accumulate<1, NT1>(X2 + 16 * w * IMG_STRIDE, X1, G2, g, c);
      acc = acc0 + acc1;
      split_delta_stored<C, DROP>(acc, a1u, d1l, ik, g, w);

      // ---- layer 1: dW1 row tile w; W1^T delta1 split over the own k-steps
#pragma unroll
      for (int r = 0; r < 4; ++r) imgD1[(4 * r + g) * IMG_STRIDE + c] = d1l[r];
      f32x4 part = z;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * w + r < M::QW) part = mfma4(F.B1[r], d1l[r], part);
      wave_lds_sync();
      dw_accumulate<1, 1>(imgD1, imgB0, G1, g, c);
      float gh[M::QH];
      split_reduce<M::QH>(PR, part, gh, lane, w);      // (c)
#pragma unroll
      for (int q = 0; q < M::QH; ++q) {
        const float th = b0[q];  // = tanh(h) wherever unit 4q + g < H
        const float dth = (4 * q + g) < C::H ? 1.0f - th * th : 0.0f;
        lam[q] = fmaf(gh[q], dth, lam[q]);
      }
    }
    if (w == 0) {
      float* out = valid ? a.lam_start + (size_t)it.r * C::H : trash;
#pragma unroll
      for (int q = 0; q < M::QH; ++q) {
        const int u = 4 * q + g;
        float* dst = u < C::H ? out + u : trash;
        *dst = lam[q];
      }
    }
  }

  BWD_STAMP(2, wall_clock64());
  BWD_STAMP(4, (unsigned long long)n_done);
  BWD_STAMP(5, (unsigned long long)n_steps_done);
  (void)n_done; (void)n_steps_done;
  // ---- flush the tiles this wave owns into the block's slab row (parameter layout)
  float* slab = a.slab + (size_t)slab_row * C::P + C::OFF_ODE;
  float *W1 = slab + NL::woff(0), *b1 = slab + NL::boff(0), *W2 = slab + NL::woff(1),
        *b2 = slab + NL::boff(1), *W3 = slab + NL::woff(2), *b3 = slab + NL::boff(2);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int uo = 16 * w + 4 * g + r;
    if (uo < M::W) {
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) {
        const int ui = 16 * nt + c;
        if (ui < M::W) W2[uo * M::W + ui] = G2[0][nt][r];
        else if (ui == M::W) b2[uo] = G2[0][nt][r];
      }
      if (c < M::IN0) W1[uo * M::IN0 + M::col0(c)] = G1[0][0][r];
      else if (c == M::IN0) b1[uo] = G1[0][0][r];
    }
    const int uh = 4 * g + r, ui = 16 * w + c;
    if (uh < C::H) {
      if (ui < M::W) W3[uh * M::W + ui] = G3[0][0][r];
      else if (ui == M::W) b3[uh] = G3[0][0][r];
    }
  }
  BWD_STAMP(3, wall_clock64());
}


// ---- kernels ------------------------------------------------------------------------------
// Pure split form (tails; plans too small to mix)
template <class C, bool DROP, bool TAIL>
__global__ void __launch_bounds__(256, 2) k_ode_fwd_split(KArgs a) {
  __shared__ __attribute__((aligned(16))) float lds_raw[OdeFwdSplitLds<C>::FLOATS];
  const int n_items = TAIL ? a.B : a.n_obs;
  // (only launched for the tails: the items of a saving forward go through the mixed kernel)
  ode_fwd_split<C, DROP, TAIL, false>(a, (lfp)lds_raw, blockIdx.x, gridDim.x, 0, (n_items + 15) / 16);
}

// Tails of a LARGE plan (round 5): one wave per tile on the scaled fragments, like the bulk of
// the mixed kernel -- the tails run BESIDE the items' forward on a busy chip, where the throughput
// per SIMD counts, not the latency of one tile (the four-wave form above: ~20 % less per SIMD)
template <class C, bool DROP>
__global__ void __launch_bounds__(64) k_ode_fwd_tails(KArgs a) {
  ode2_fwd_single<C, DROP, true, false>(a, threadIdx.x, blockIdx.x, gridDim.x, 0, (a.B + 15) / 16);
}

// Mixed form.  Tiles are sorted by length; the split point T (k_split_point, on the device)
// hands the T longest tiles to the first `ns` blocks, which run them with four waves per tile
// (low latency per Euler step), and the rest to the other blocks, whose waves each run a tile
// of their own (higher throughput per SIMD).  T balances the two groups' finishing times,
// so the long segments no longer set the kernel time and the bulk still runs at the
// single-wave kernels' rate.
// (ENC: the NJODE_ENC_FUSED=1 form -- a kernel of its own, so that the default one keeps its
// registers and its LDS footprint)
// (PLAN: the launch carries the NEXT batch's plan in front of its own blocks -- njode_plan.h; a
// kernel of its own, so that launches without a job keep their argument list and their registers)
template <class C, bool ENC, bool PLAN> struct OdeFwdMixedLds {
  static constexpr int OWN = OdeFwdSplitLds<C>::FLOATS + (ENC ? EncFwdLds<C>::FLOATS : 0);
  static constexpr int FLOATS = (PLAN && PLAN_LDS_INTS > OWN) ? PLAN_LDS_INTS : OWN;
};
template <class C, bool DROP, bool ENC, bool PLAN>
__device__ __forceinline__ void ode_fwd_mixed_body(const KArgs& a, lfp lds_raw, int bid, int nblocks) {
  const int n_tiles = (a.n_obs + 15) / 16, ns = a.n_split_fwd;
  const int T = (int)a.base_s[a.K + 2];
  const bool save = a.save_traj != 0;   // wave-uniform: a training forward stores checkpoints
                                         // and activations, an evaluation forward nothing
  // (the backward's tile queue starts from zero: its last block clears it again, this covers the
  // very first launch on a fresh workspace)
  if (save && bid == 0 && threadIdx.x == 0) { a.tile_q[0] = 0; a.tile_q[1] = 0; a.tile_q[2] = 0; }
  if (bid < ns) {
    if (save) ode_fwd_split<C, DROP, false, true>(a, lds_raw, bid, ns, 0, T);
    else ode_fwd_split<C, DROP, false, false>(a, lds_raw, bid, ns, 0, T);
  } else {
    const int wave = (bid - ns) * 4 + uniform(threadIdx.x >> 6);
    // one-wave role on the scaled fragments (njode_ode2.h): same masks, same values to rounding
    const int nw = (nblocks - ns) * 4;
    if constexpr (ENC) {   // the wave evaluates the encoder at the head of every item
      lfp enc_img = lds_raw + OdeFwdSplitLds<C>::FLOATS;
      EncFwdLds<C>::stage(enc_img, a.frag_enc, threadIdx.x, 256);
      __syncthreads();
      if (save) ode2_fwd_single<C, DROP, false, true, true>(a, threadIdx.x & 63, wave, nw, T, n_tiles, enc_img);
      else ode2_fwd_single<C, DROP, false, false, true>(a, threadIdx.x & 63, wave, nw, T, n_tiles, enc_img);
    } else {
      if (save) ode2_fwd_single<C, DROP, false, true>(a, threadIdx.x & 63, wave, nw, T, n_tiles);
      else ode2_fwd_single<C, DROP, false, false>(a, threadIdx.x & 63, wave, nw, T, n_tiles);
    }
  }
}
template <class C, bool DROP, bool ENC = false>
__global__ void __launch_bounds__(256, 2) k_ode_fwd_mixed(KArgs a) {
  // (ENC: behind the four-wave role's exchange images, the encoder's forward fragments)
  __shared__ __attribute__((aligned(16))) float lds_raw[OdeFwdMixedLds<C, ENC, false>::FLOATS];
  ode_fwd_mixed_body<C, DROP, ENC, false>(a, (lfp)lds_raw, blockIdx.x, gridDim.x);
}
template <class C, bool DROP, bool ENC = false>
__global__ void __launch_bounds__(256, 2) k_ode_fwd_mixed_plan(KArgs a, PlanJob job) {
  __shared__ __attribute__((aligned(16))) float lds_raw[OdeFwdMixedLds<C, ENC, true>::FLOATS];
  if ((int)blockIdx.x < job.P) {
    plan_grid_body(job, blockIdx.x, (int*)lds_raw);
    return;
  }
  ode_fwd_mixed_body<C, DROP, ENC, true>(a, (lfp)lds_raw, (int)blockIdx.x - job.P, (int)gridDim.x - job.P);
}
template <class C> struct OdeBwdMixedLds {
  static constexpr int A = OdeBwdActLds<C>::FLOATS, B = OdeBwdSplitLds<C>::FLOATS;
  static constexpr int FLOATS = A > B ? A : B;
};
// one slab row per block
// (QUEUE: the NJODE_BWD_QUEUE=1 form -- a kernel of its own: in one kernel with the static rounds its
// extra live values pushed the register allocation of BOTH over the edge: 256 VGPRs + 75 spilled)
template <class C, bool DROP, bool QUEUE = false>
__global__ void __launch_bounds__(256, 2) k_ode_bwd_mixed(KArgs a) {
  __shared__ __attribute__((aligned(16))) float lds_raw[OdeBwdMixedLds<C>::FLOATS];
  const int n_tiles = (a.n_obs + 15) / 16, ns = a.n_split_blocks;
  const int T = (int)a.base_s[a.K + 1];
  // both roles read the forward's step records (njode_ode2.h): the mixed kernels only run with
  // a saved forward, which always has them
#ifdef NJ_BWD_STAMPS
  BWD_STAMP(0, wall_clock64());
  BWD_STAMP(6, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4 /* HW_ID */) |
                   ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20 /* XCC_ID */) << 32));
  BWD_STAMP(7, (unsigned long long)((int)blockIdx.x < ns ? 1 : 0) | ((unsigned long long)T << 8) |
                   ((unsigned long long)gridDim.x << 32));
#endif
  if constexpr (QUEUE) {   // tile queue, persistent blocks
    if ((int)blockIdx.x < ns) {
      ode3_bwd_split<C, DROP, true>(a, (lfp)lds_raw, blockIdx.x, ns, 0, T, blockIdx.x, n_tiles);
    } else {
      const int wave = ((int)blockIdx.x - ns) * 4 + uniform(threadIdx.x >> 6);
      ode3_bwd_single<C, DROP, true>(a, (lfp)lds_raw, wave, ((int)gridDim.x - ns) * 4, T, n_tiles, blockIdx.x);
    }
    queue_block_done(a.tile_q, gridDim.x);
  } else {
    if ((int)blockIdx.x < ns) {
      ode3_bwd_split<C, DROP>(a, (lfp)lds_raw, blockIdx.x, ns, 0, T, blockIdx.x);
    } else {
      const int wave = ((int)blockIdx.x - ns) * 4 + uniform(threadIdx.x >> 6);
      ode3_bwd_single<C, DROP>(a, (lfp)lds_raw, wave, ((int)gridDim.x - ns) * 4, T, n_tiles, blockIdx.x);
    }
  }
}

}  // namespace njode
