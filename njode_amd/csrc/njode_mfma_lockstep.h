// njode_mfma_lockstep.h -- the lockstep plan's forward pass on the f32 matrix cores:
// 16 paths per wave over the shared grid (all modes: masked self-imputation, prediction
// paths for get_pred / evaluate, schedules with a tail).  Same D-layout conventions as
// njode_mfma.h; the three networks' forward A-fragments live once per 256-thread block in
// LDS.  Per-chain vectors that several lane groups need in full (last observation, mask,
// readout) are kept as per-lane arrays; the readout is gathered through an LDS image.
#pragma once
#include "njode_mfma_rows.h"

#include <type_traits>

namespace njode {

// b[q] = f(unit 4q + g), with the unit index a compile-time constant inside f
template <int Q, int NQ, class F> NJ_DEV void fill_units(float (&b)[NQ], int g, F f) {
  if constexpr (Q < NQ) {
    const float e0 = f(std::integral_constant<int, 4 * Q + 0>{});
    const float e1 = f(std::integral_constant<int, 4 * Q + 1>{});
    const float e2 = f(std::integral_constant<int, 4 * Q + 2>{});
    const float e3 = f(std::integral_constant<int, 4 * Q + 3>{});
    b[Q] = g == 0 ? e0 : (g == 1 ? e1 : (g == 2 ? e2 : e3));
    fill_units<Q + 1, NQ>(b, g, f);
  }
}

// forward fragments of the ODE network in LDS (MF<C> table, vectors [0, NFWD))
template <class C> struct OdeFwdLds {
  using M = MF<C>;
  lfp base, cur;
  static NJ_DEV void stage(lfp img, const float* frag, int tid, int nthreads) {
    for (int i = tid; i < M::NFWD * 64; i += nthreads) img[i] = frag[i];
  }
  NJ_DEV void init(lfp img, int lane) { base = img + lane; cur = base; }
  NJ_DEV void begin() {
    unsigned v = (unsigned)(unsigned long long)base;
    asm volatile("" : "+v"(v));
    cur = (lfp)(unsigned long long)v;
  }
  NJ_DEV float a1(int mt, int q) const { return cur[(M::F1 + mt * M::Q0 + q) * 64]; }
  NJ_DEV float a2(int mt, int q) const { return cur[(M::F2 + mt * M::Q1 + q) * 64]; }
  NJ_DEV float a3(int mt, int q) const { return cur[(M::F3 + mt * M::Q1 + q) * 64]; }
};

// store / load a D-layout vector to / from a row of N floats
template <int NQ, int N> NJ_DEV void store_units(float* row, const float (&v)[NQ], int g, float* trash) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int u = 4 * q + g;
    float* dst = u < N ? row + u : trash;
    *dst = v[q];
  }
}

template <class C, bool DROP>
__global__ void __launch_bounds__(128) k_paths_fwd_mfma(KArgs a) {
  constexpr int NWV = 2, NT = NWV * 64;   // 2 waves = 32 paths per block (LDS budget)
  using M = MF<C>;
  using ES = typename EncS<C>::type;
  using DS = typename DecS<C>::type;
  constexpr int D = C::D, H = C::H, DO = C::DO;
  constexpr int FR = M::NFWD + ES::NFWD + DS::NFWD;           // fragment vectors in LDS
  __shared__ __attribute__((aligned(16))) float lds_raw[FR * 64 + NWV * 2 * IMG_FLOATS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  lfp f_ode = (lfp)lds_raw, f_enc = f_ode + M::NFWD * 64, f_dec = f_enc + ES::NFWD * 64;
  lfp img_h = (lfp)lds_raw + FR * 64 + wv * 2 * IMG_FLOATS, img_o = img_h + IMG_FLOATS;
  OdeFwdLds<C>::stage(f_ode, a.frag, threadIdx.x, NT);
  for (int i = threadIdx.x; i < ES::NFWD * 64; i += NT) f_enc[i] = a.frag_enc[i];
  for (int i = threadIdx.x; i < DS::NFWD * 64; i += NT) f_dec[i] = a.frag_dec[i];
  for (int i = threadIdx.x; i < NWV * 2 * IMG_FLOATS; i += NT) lds_raw[FR * 64 + i] = 0.0f;
  __syncthreads();
  OdeFwdLds<C> Fo;
  LdsFrags<ES> Fe;
  LdsFrags<DS> Fd;
  Fo.init(f_ode, lane);
  Fe.init(f_enc, lane);
  Fd.init(f_dec, lane);

  const bool PATH = a.want_path != 0, LOSS = a.want_loss != 0, SAVE = a.save_traj != 0;
  const int tile = blockIdx.x * NWV + wv;
  const int b0i = tile * 16 + c;
  const bool valid = b0i < a.B;
  const int b = valid ? b0i : a.B - 1;
  const unsigned long long gid = a.gid0 + b;
  float* const trash = a.trash + threadIdx.x * (H > D ? H : D);
  const int __attribute__((address_space(4)))* kjump =
      (const int __attribute__((address_space(4)))*)(unsigned long long)a.k_jump;
  const cfp sdt = as_cfp(a.step_dt), stt = as_cfp(a.step_t), tf = as_cfp(a.time_f32);

  // ---- network evaluations on D-layout state -------------------------------------------------
  auto keep2 = [&](uint32_t tkey, uint32_t net, uint32_t& k1, uint32_t& k2) {
    row_keep_bits<DROP>(a, gid, tkey, net, g, 16, k1, k2);
  };
  // y[DO] (full, per lane) = readout(h)
  auto readout_full = [&](const float (&hq)[M::QH], uint32_t tkey, uint32_t net, float (&y)[DO]) {
    float b0d[DS::Q0], a1[DS::Q1], a2[DS::Q1], o[DS::QO];
    f32x4 out[DS::MTO];
#pragma unroll
    for (int q = 0; q < DS::Q0; ++q) {
      const int u = 4 * q + g;
      const float th = q < M::QH ? tanh_f(hq[q < M::QH ? q : 0]) : 0.0f;
      b0d[q] = u < H ? th : (u == H ? 1.0f : 0.0f);
    }
    img_write<M::QH>(img_h, hq, g, c);
    uint32_t k1, k2;
    keep2(tkey, net, k1, k2);
    mnet_fwd<DS, C::ACT, DROP>(Fd, b0d, a1, a2, out, k1, k2, a.dc.inv_keep, g);
#pragma unroll
    for (int q = 0; q < DS::QO; ++q) o[q] = out[q / 4][q % 4];
    img_write<DS::QO>(img_o, o, g, c);
    wave_lds_sync();
#pragma unroll
    for (int u = 0; u < DO; ++u) {
      float v = img_o[u * IMG_STRIDE + c];
      if constexpr (C::DEC_CASE == 1) {
        v += img_h[(u % H) * IMG_STRIDE + c];
      } else if constexpr (C::DEC_CASE == 2) {
        constexpr int mult = H / DO;
        float s = 0.0f;
#pragma unroll
        for (int cc = 0; cc < mult; ++cc) s += img_h[(cc * DO + u) * IMG_STRIDE + c];
        v += s * (1.0f / mult);
      }
      y[u] = v;
    }
    wave_lds_sync();
  };
  // hq (D-layout) = encoder(xin, mask)
  auto encode_full = [&](const float (&xin)[D], const float (&mask)[D], uint32_t tkey,
                         float (&hq)[M::QH]) {
    float b0e[ES::Q0], a1[ES::Q1], a2[ES::Q1];
    f32x4 out[ES::MTO];
    fill_units<0, ES::Q0>(b0e, g, [&](auto U) {
      constexpr int u = decltype(U)::value;
      if constexpr (u < D) return tanh_f(xin[u]);
      else if constexpr (C::MASKED && u < 2 * D) return mask[u - D];
      else if constexpr (u == C::ENC_IN) return 1.0f;
      else return 0.0f;
    });
    uint32_t k1, k2;
    keep2(tkey, NET_ENC, k1, k2);
    mnet_fwd<ES, C::ACT, DROP>(Fe, b0e, a1, a2, out, k1, k2, a.dc.inv_keep, g);
    float res[M::QH];
    fill_units<0, M::QH>(res, g, [&](auto U) {
      constexpr int u = decltype(U)::value;
      if constexpr (u >= H) {
        return 0.0f;
      } else if constexpr (C::ENC_CASE == 1) {
        return xin[u % D];
      } else if constexpr (C::ENC_CASE == 2) {
        constexpr int mult = D / H;
        float s = 0.0f;
#pragma unroll
        for (int cc = 0; cc < mult; ++cc) s += xin[cc * H + u];
        return s * (1.0f / mult);
      } else {
        return 0.0f;
      }
    });
#pragma unroll
    for (int q = 0; q < M::QH; ++q) hq[q] = out[q / 4][q % 4] + res[q];
  };

  // ---- initial state --------------------------------------------------------------------------
  float xl[D], mask[D], tx[D], h[M::QH], y[DO];
  load_vec(a.start_X + (size_t)b * D, xl);
#pragma unroll
  for (int i = 0; i < D; ++i) { mask[i] = 0.0f; tx[i] = tanh_f(xl[i]); }
#pragma unroll
  for (int i = 0; i < DO; ++i) y[i] = 0.0f;
  encode_full(xl, mask, TKEY_START, h);
  float tau = 0.0f, loss_acc = 0.0f;
  int cur = a.first_j[b];
  int next_i = a.n_obs > 0 ? a.t_of_row[a.row_by_path[cur >= 0 ? cur : 0]] : 0;
  next_i = cur >= 0 ? next_i : 0x7fffffff;
  int src = -1, row = 0;

  auto write_row = [&]() {
    if (PATH) {
      store_units<M::QH, H>(valid ? a.path_h + ((size_t)row * a.B + b) * H : trash, h, g, trash);
      if (g == 0) store_vec(valid ? a.path_y + ((size_t)row * a.B + b) * DO : trash, y);
      ++row;
    }
  };
  if (PATH) readout_full(h, TKEY_START - 1, NET_DEC_ROW, y);
  write_row();

  int i = 0;
  for (int k = 0;; ++k) {
    while (i < a.n_times && kjump[i] == k) {
      const bool has = valid && next_i == i;
      if (__any(has)) {  // wave-uniform
        const int r = has ? a.row_by_path[cur >= 0 ? cur : 0] : 0;
        float ybj[DO], yn[DO], x[D], m[D], xin[D], hn[M::QH];
        if (SAVE) store_units<M::QH, H>(has ? a.h_end + (size_t)r * H : trash, h, g, trash);
        readout_full(h, (uint32_t)k, NET_DEC_BJ, ybj);
        load_vec(a.X + (size_t)r * D, x);
        if constexpr (C::MASKED) {
          load_vec(a.M + (size_t)r * D, m);
#pragma unroll
          for (int q = 0; q < D; ++q) xin[q] = x[q] * m[q] + (1.0f - m[q]) * ybj[q];
        } else {
#pragma unroll
          for (int q = 0; q < D; ++q) { xin[q] = x[q]; m[q] = 1.0f; }
        }
        encode_full(xin, m, (uint32_t)k, hn);
        readout_full(hn, (uint32_t)k, NET_DEC, yn);
        if (LOSS) {
          float dy[DO], dybj[DO];
          const float scale = a.inv_batch * __builtin_amdgcn_rcpf((float)a.n_obs_ot[b]);
          const float term = loss_row<C>(x, m, yn, ybj, a.weight, a.loss_easy, scale, dy, dybj);
          loss_acc += has ? term : 0.0f;
        }
        if (SAVE && g == 0) {
          store_vec(has ? a.y_row + (size_t)r * DO : trash, yn);
          store_vec(has ? a.ybj_row + (size_t)r * DO : trash, ybj);
        }
        // commit for the chains that have an observation (models.py:463-489)
#pragma unroll
        for (int q = 0; q < M::QH; ++q) h[q] = has ? hn[q] : h[q];
#pragma unroll
        for (int q = 0; q < DO; ++q) y[q] = has ? yn[q] : y[q];
#pragma unroll
        for (int q = 0; q < D; ++q) {
          const float nt = tanh_f(C::MASKED ? yn[q] : x[q]);
          tx[q] = has ? nt : tx[q];
        }
        const float tnew = tf[i];
        tau = has ? tnew : tau;
        src = has ? r : src;
        const int cn = cur + 1;
        const int cc = cn < a.n_obs ? cn : 0;
        const int nt_ = a.t_of_row[a.row_by_path[cc]];
        const int nexti2 = (cn < a.n_obs && a.path_sorted[cc] == b) ? nt_ : 0x7fffffff;
        cur = has ? cn : cur;
        next_i = has ? nexti2 : next_i;
      }
      write_row();
      ++i;
    }
    if (SAVE) {
      store_units<M::QH, H>(valid ? a.ltraj + ((size_t)k * a.B + b) * H : trash, h, g, trash);
      if (valid && g == 0 && k < a.K) a.src_row[(size_t)k * a.B + b] = src;
    }
    if (k >= a.K) break;
    {
      const float dt = sdt[k], t = stt[k];
      float b0[M::Q0], a1[M::Q1], a2[M::Q1];
      in0_fill<C, 0>(b0, h, tx, tau, t - tau, g);
      uint32_t k1 = 0, k2 = 0;
      if constexpr (DROP) {
        uint32_t st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1),
                                 (uint32_t)k, NET_ODE);
        k1 = keep_bits<M::Q1>(st, a.dc.thr16);
        k2 = keep_bits<M::Q1>(st, a.dc.thr16);
      }
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      f32x4 acc[M::MT1], acch[M::MTH];
      Fo.begin();
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = z;
#pragma unroll
      for (int q = 0; q < M::Q0; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(Fo.a1(mt, q), b0[q], acc[mt]);
      hidden_from_acc<C, DROP>(acc, a1, k1, a.dc.inv_keep, g);
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = z;
#pragma unroll
      for (int q = 0; q < M::Q1; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(Fo.a2(mt, q), a1[q], acc[mt]);
      hidden_from_acc<C, DROP>(acc, a2, k2, a.dc.inv_keep, g);
#pragma unroll
      for (int mt = 0; mt < M::MTH; ++mt) acch[mt] = z;
#pragma unroll
      for (int q = 0; q < M::Q1; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MTH; ++mt) acch[mt] = mfma4(Fo.a3(mt, q), a2[q], acch[mt]);
#pragma unroll
      for (int q = 0; q < M::QH; ++q) h[q] = fmaf(dt, acch[q / 4][q % 4], h[q]);
      if (PATH) readout_full(h, 0x80000000u + (uint32_t)k, NET_DEC_ROW, y);
      write_row();
    }
  }
  store_units<M::QH, H>(valid ? a.hT + (size_t)b * H : trash, h, g, trash);
  if (LOSS && valid && g == 0) a.loss_terms[b] = loss_acc;
}


// =====================================================================================
// Adjoint sweep of the lockstep plan on the matrix cores (pass 1 of its backward; the
// weight-gradient kernels of njode_lockstep_bwd.h are pass 2).  16 paths per wave, events
// in reverse.  Only W^T products are needed, so no gradient tiles: the A-fragments of all
// three networks are streamed from L2 (they exceed the LDS for the 41-dimensional shapes).
// =====================================================================================

// fragments read straight from global memory; laundered per use so the loop-invariant
// loads are not hoisted into hundreds of registers
template <class S> struct GFrags {
  const float* base;
  const float* cur;
  NJ_DEV void init(const float* frag, int lane) { base = frag + lane; cur = base; }
  NJ_DEV void begin() {
    unsigned long long v = (unsigned long long)base;
    asm volatile("" : "+v"(v));
    cur = (const float*)v;
  }
  NJ_DEV float a1(int mt, int q) const { return cur[(S::F1 + mt * S::Q0 + q) * 64]; }
  NJ_DEV float a2(int mt, int q) const { return cur[(S::F2 + mt * S::Q1 + q) * 64]; }
  NJ_DEV float a3(int mt, int q) const { return cur[(S::F3 + mt * S::Q1 + q) * 64]; }
  NJ_DEV float b3(int mt, int q) const { return cur[(S::B3 + mt * S::QO + q) * 64]; }
  NJ_DEV float b2(int mt, int q) const { return cur[(S::B2 + mt * S::QW + q) * 64]; }
  NJ_DEV float b1(int mt, int q) const { return cur[(S::B1 + mt * S::QW + q) * 64]; }
};
template <class C> struct OdeGFrags {
  using M = MF<C>;
  const float* base;
  const float* cur;
  NJ_DEV void init(const float* frag, int lane) { base = frag + lane; cur = base; }
  NJ_DEV void begin() {
    unsigned long long v = (unsigned long long)base;
    asm volatile("" : "+v"(v));
    cur = (const float*)v;
  }
  NJ_DEV float a1(int mt, int q) const { return cur[(M::F1 + mt * M::Q0 + q) * 64]; }
  NJ_DEV float a2(int mt, int q) const { return cur[(M::F2 + mt * M::Q1 + q) * 64]; }
  NJ_DEV float b3(int mt, int q) const { return cur[(M::B3 + mt * M::QH + q) * 64]; }
  NJ_DEV float b2(int mt, int q) const { return cur[(M::B2 + mt * M::QW + q) * 64]; }
  NJ_DEV float b1(int mt, int q) const { return cur[(M::B1 + mt * M::QW + q) * 64]; }
};

// the same in LDS without W3 (the sweep only needs the hidden activations and W^T products)
template <class S> struct AdjLdsFrags {
  static constexpr int SKIP = S::NFWD - S::F3;
  static constexpr int NVEC = S::NALL - SKIP;
  lfp base, cur;
  static NJ_DEV void stage(lfp img, const float* frag, int tid, int nthreads) {
    for (int i = tid; i < S::F3 * 64; i += nthreads) img[i] = frag[i];
    for (int i = tid; i < (S::NALL - S::NFWD) * 64; i += nthreads)
      img[S::F3 * 64 + i] = frag[S::NFWD * 64 + i];
  }
  NJ_DEV void init(lfp img, int lane) { base = img + lane; cur = base; }
  NJ_DEV void begin() {
    unsigned v = (unsigned)(unsigned long long)base;
    asm volatile("" : "+v"(v));
    cur = (lfp)(unsigned long long)v;
  }
  NJ_DEV float a1(int mt, int q) const { return cur[(S::F1 + mt * S::Q0 + q) * 64]; }
  NJ_DEV float a2(int mt, int q) const { return cur[(S::F2 + mt * S::Q1 + q) * 64]; }
  NJ_DEV float b3(int mt, int q) const { return cur[(S::B3 - SKIP + mt * S::QO + q) * 64]; }
  NJ_DEV float b2(int mt, int q) const { return cur[(S::B2 - SKIP + mt * S::QW + q) * 64]; }
  NJ_DEV float b1(int mt, int q) const { return cur[(S::B1 - SKIP + mt * S::QW + q) * 64]; }
};

// hidden activations of one evaluation (layers 1 and 2 of mnet_fwd)
template <class S, int ACT, bool DROP, class FP>
NJ_DEV void mnet_hidden(FP& F, const float (&b0)[S::Q0], float (&a1)[S::Q1], float (&a2)[S::Q1],
                        uint32_t k1, uint32_t k2, float inv_keep, int g) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[S::MT1];
  F.begin();
#pragma unroll
  for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = z;
#pragma unroll
  for (int q = 0; q < S::Q0; ++q)
#pragma unroll
    for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = mfma4(F.a1(mt, q), b0[q], acc[mt]);
  hidden_from_acc_g<S::MT1, S::Q1, S::W, ACT, DROP>(acc, a1, k1, inv_keep, g);
#pragma unroll
  for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = z;
#pragma unroll
  for (int q = 0; q < S::Q1; ++q)
#pragma unroll
    for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = mfma4(F.a2(mt, q), a1[q], acc[mt]);
  hidden_from_acc_g<S::MT1, S::Q1, S::W, ACT, DROP>(acc, a2, k2, inv_keep, g);
}

// transposed products of one evaluation: d/d (first NT input tiles), no weight gradients
template <class S, int ACT, bool DROP, int NT, class FP>
NJ_DEV void mnet_adj(FP& Bf, const float (&dout)[S::QO], const float (&a1)[S::Q1],
                     const float (&a2)[S::Q1], uint32_t k1, uint32_t k2, float inv_keep,
                     float keepf, f32x4 (&din)[NT]) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[S::MT1];
  Bf.begin();
#pragma unroll
  for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = z;
#pragma unroll
  for (int q = 0; q < S::QO; ++q)
#pragma unroll
    for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = mfma4(Bf.b3(mt, q), dout[q], acc[mt]);
  float d2[S::QW], d1[S::QW];
  hidden_delta_g<S::MT1, S::Q1, S::QW, ACT, DROP>(acc, a2, d2, k2, inv_keep, keepf);
#pragma unroll
  for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = z;
#pragma unroll
  for (int q = 0; q < S::QW; ++q)
#pragma unroll
    for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = mfma4(Bf.b2(mt, q), d2[q], acc[mt]);
  hidden_delta_g<S::MT1, S::Q1, S::QW, ACT, DROP>(acc, a1, d1, k1, inv_keep, keepf);
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) din[mt] = z;
#pragma unroll
  for (int q = 0; q < S::QW; ++q)
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) din[mt] = mfma4(Bf.b1(mt, q), d1[q], din[mt]);
}

// gather a D-layout row [N] of floats into registers
template <int NQ, int N> NJ_DEV void load_units(const float* row, float (&v)[NQ], int g) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int u = 4 * q + g;
    const float t = row[u < N ? u : 0];
    v[q] = u < N ? t : 0.0f;
  }
}

template <class C, bool DROP>
__global__ void __launch_bounds__(64, 1) k_paths_bwd_adj_mfma(KArgs a) {
  using M = MF<C>;
  using ES = typename EncS<C>::type;
  using DS = typename DecS<C>::type;
  constexpr int D = C::D, H = C::H, DO = C::DO;
  constexpr int MTIH = (H + 15) / 16;   // readout input tiles
  constexpr int MTID = (D + 15) / 16;   // encoder input tiles that carry x
  static_assert(!C::MASKED || C::ENC_CASE == 0 || (C::ENC_CASE == 1 && D == H),
                "masked sweep: the encoder's identity path must map unit to unit");
  // the ODE network's fragments are read every step, the readout's at every jump: both live
  // in LDS when they fit beside each other; the encoder's (masked jumps only) stream from L2
  using OL = OdeLdsFrags<C>;
  using DL = AdjLdsFrags<DS>;
  constexpr int LDS_MAX = 160 * 1024 / 4 - IMG_FLOATS;
  constexpr bool ODE_LDS = OL::NVEC * 64 <= LDS_MAX;
  constexpr bool DEC_LDS = ODE_LDS && (OL::NVEC + DL::NVEC) * 64 <= LDS_MAX;
  constexpr int ODE_FL = ODE_LDS ? OL::NVEC * 64 : 0, DEC_FL = DEC_LDS ? DL::NVEC * 64 : 0;
  __shared__ __attribute__((aligned(16))) float lds_raw[IMG_FLOATS + ODE_FL + DEC_FL];
  lfp img = (lfp)lds_raw;
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  for (int i = lane; i < IMG_FLOATS; i += 64) lds_raw[i] = 0.0f;
  std::conditional_t<ODE_LDS, OL, OdeGFrags<C>> Fo;
  std::conditional_t<DEC_LDS, DL, GFrags<DS>> Fd;
  GFrags<ES> Fe;
  if constexpr (ODE_LDS) {
    OL::stage(img + IMG_FLOATS, a.frag, lane, 64);
    Fo.init(img + IMG_FLOATS, lane);
  } else {
    Fo.init(a.frag, lane);
  }
  if constexpr (DEC_LDS) {
    DL::stage(img + IMG_FLOATS + ODE_FL, a.frag_dec, lane, 64);
    Fd.init(img + IMG_FLOATS + ODE_FL, lane);
  } else {
    Fd.init(a.frag_dec, lane);
  }
  Fe.init(a.frag_enc, lane);
  wave_lds_sync();

  const int b0i = blockIdx.x * 16 + c;
  const bool valid = b0i < a.B;
  const int b = valid ? b0i : a.B - 1;
  const unsigned long long gid = a.gid0 + b;
  float* const trash = a.trash + lane * (H > D ? H : D);
  const int __attribute__((address_space(4)))* kjump =
      (const int __attribute__((address_space(4)))*)(unsigned long long)a.k_jump;
  const cfp sdt = as_cfp(a.step_dt), stt = as_cfp(a.step_t);

  float lam_h[M::QH], lx[M::Q0], tx[D];
  // (the adjoint of the final state: the upstream gradient of hT, if the caller has one)
#pragma unroll
  for (int q = 0; q < M::QH; ++q) {
    const int u = 4 * q + g;
    lam_h[q] = (a.g_hT && valid && u < H) ? a.g_hT[(size_t)b * H + (u < H ? u : 0)] : 0.0f;
  }
#pragma unroll
  for (int q = 0; q < M::Q0; ++q) lx[q] = 0.0f;
  int src = a.last_row[b];
  int src_i = src >= 0 ? a.t_of_row[src] : -1;
  float tau = 0.0f;
  // last_X / tau of the segment that follows row `src` (or the start value)
  auto load_source = [&](int srow) {
    const int sv = srow >= 0 ? srow : 0;
    const float* xp = srow >= 0 ? (C::MASKED ? a.y_row + (size_t)sv * DO : a.X + (size_t)sv * D)
                                : a.start_X + (size_t)b * D;
#pragma unroll
    for (int q = 0; q < D; ++q) tx[q] = tanh_f(xp[q]);
    const float tsrc = a.n_obs > 0 ? a.time_f32[a.t_of_row[sv]] : 0.0f;
    tau = srow >= 0 ? tsrc : 0.0f;
  };
  load_source(src);

  // D-layout vector -> full per-lane vector (units [U0, U0 + N)) through the LDS image
  auto to_full = [&](auto NQc, auto U0c, auto Nc, const float* v, float* out) {
    constexpr int NQ = decltype(NQc)::value, U0 = decltype(U0c)::value, N = decltype(Nc)::value;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int u = 4 * q + g - U0;
      if (u >= 0 && u < N && u < IMG_ROWS) img[u * IMG_STRIDE + c] = v[q];
    }
    wave_lds_sync();
#pragma unroll
    for (int u = 0; u < N; ++u) out[u] = img[u * IMG_STRIDE + c];
    wave_lds_sync();
  };
  auto dec_adj = [&](const float (&hq)[M::QH], const float (&dyf)[DO], uint32_t tkey, uint32_t net,
                     float (&dh)[M::QH]) {
    float b0d[DS::Q0], a1[DS::Q1], a2[DS::Q1], dq[DS::QO];
    f32x4 din[MTIH];
#pragma unroll
    for (int q = 0; q < DS::Q0; ++q) {
      const int u = 4 * q + g;
      const float th = q < M::QH ? tanh_f(hq[q < M::QH ? q : 0]) : 0.0f;
      b0d[q] = u < H ? th : (u == H ? 1.0f : 0.0f);
    }
    uint32_t k1, k2;
    row_keep_bits<DROP>(a, gid, tkey, net, g, 16, k1, k2);
    mnet_hidden<DS, C::ACT, DROP>(Fd, b0d, a1, a2, k1, k2, a.dc.inv_keep, g);
    fill_by_group<DS::QO, DO>(dq, dyf, g);
    mnet_adj<DS, C::ACT, DROP, MTIH>(Fd, dq, a1, a2, k1, k2, a.dc.inv_keep, a.keep, din);
    float res[M::QH];
    fill_units<0, M::QH>(res, g, [&](auto U) {
      constexpr int u = decltype(U)::value;
      if constexpr (u >= H) {
        return 0.0f;
      } else if constexpr (C::DEC_CASE == 1) {
        float s_ = 0.0f;
#pragma unroll
        for (int j = u; j < DO; j += H) s_ += dyf[j];
        return s_;
      } else if constexpr (C::DEC_CASE == 2) {
        return dyf[u % DO] * (1.0f / (H / DO));
      } else {
        return 0.0f;
      }
    });
#pragma unroll
    for (int q = 0; q < M::QH; ++q) {
      const float th = b0d[q < DS::Q0 ? q : 0];
      dh[q] = (4 * q + g < H) ? din[q / 4][q % 4] * (1.0f - th * th) + res[q] : 0.0f;
    }
  };

  int i = a.n_times - 1;
  for (int k = a.K; k >= 0; --k) {
    if (k < a.K) {
      // ---- reverse Euler step k
      float h[M::QH], b0[M::Q0], a1[M::Q1], a2[M::Q1], d3[M::QH];
      load_units<M::QH, H>(a.ltraj + ((size_t)k * a.B + b) * H, h, g);
      const float dt = sdt[k], t = stt[k];
      in0_fill<C, 0>(b0, h, tx, tau, t - tau, g);
      uint32_t k1 = 0, k2 = 0;
      if constexpr (DROP) {
        uint32_t st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1),
                                 (uint32_t)k, NET_ODE);
        k1 = keep_bits<M::Q1>(st, a.dc.thr16);
        k2 = keep_bits<M::Q1>(st, a.dc.thr16);
      }
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      f32x4 acc[M::MT1], din[M::MTB1];
      Fo.begin();
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = z;
#pragma unroll
      for (int q = 0; q < M::Q0; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(Fo.a1(mt, q), b0[q], acc[mt]);
      hidden_from_acc<C, DROP>(acc, a1, k1, a.dc.inv_keep, g);
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = z;
#pragma unroll
      for (int q = 0; q < M::Q1; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(Fo.a2(mt, q), a1[q], acc[mt]);
      hidden_from_acc<C, DROP>(acc, a2, k2, a.dc.inv_keep, g);
      store_units<M::QH, H>(valid ? a.lam_traj + ((size_t)k * a.B + b) * H : trash, lam_h, g, trash);
#pragma unroll
      for (int q = 0; q < M::QH; ++q) d3[q] = dt * lam_h[q];
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = z;
#pragma unroll
      for (int q = 0; q < M::QH; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(Fo.b3(mt, q), d3[q], acc[mt]);
      float d2[M::QW], d1[M::QW];
      hidden_delta<C, DROP>(acc, a2, d2, k2, a.dc.inv_keep, a.keep);
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = z;
#pragma unroll
      for (int q = 0; q < M::QW; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(Fo.b2(mt, q), d2[q], acc[mt]);
      hidden_delta<C, DROP>(acc, a1, d1, k1, a.dc.inv_keep, a.keep);
#pragma unroll
      for (int mt = 0; mt < M::MTB1; ++mt) din[mt] = z;
#pragma unroll
      for (int q = 0; q < M::QW; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MTB1; ++mt) din[mt] = mfma4(Fo.b1(mt, q), d1[q], din[mt]);
      // in0 units: [h (H), x (D), ...]; both are tanh'd inputs
#pragma unroll
      for (int q = 0; q < 4 * M::MTB1 && q < M::Q0; ++q) {
        const int u = 4 * q + g;
        const float th = b0[q];
        const float gq = din[q / 4][q % 4] * (1.0f - th * th);
        if (q < M::QH) lam_h[q < M::QH ? q : 0] += u < H ? gq : 0.0f;
        if constexpr (C::MASKED) lx[q] += (u >= H && u < H + D) ? gq : 0.0f;
      }
    }
    // ---- reverse the jump applied right before step k
    while (i >= 0 && kjump[i] == k) {
      const bool has = valid && src >= 0 && src_i == i;
      if (__any(has)) {
        const int r = has ? src : 0;
        float hn[M::QH], hp[M::QH], x[D], m[D], y[DO], ybj[DO], dy[DO], dybj[DO];
        float dh[M::QH], lam_hn[M::QH], lam_new[M::QH];
        load_units<M::QH, H>(a.ltraj + ((size_t)k * a.B + b) * H, hn, g);
        load_units<M::QH, H>(a.h_end + (size_t)r * H, hp, g);
        load_vec(a.X + (size_t)r * D, x);
        if constexpr (C::MASKED) {
          load_vec(a.M + (size_t)r * D, m);
        } else {
#pragma unroll
          for (int q = 0; q < D; ++q) m[q] = 1.0f;
        }
        load_vec(a.y_row + (size_t)r * DO, y);
        load_vec(a.ybj_row + (size_t)r * DO, ybj);
        const float scale = a.inv_batch * __builtin_amdgcn_rcpf((float)a.n_obs_ot[b]);
        loss_row<C>(x, m, y, ybj, a.weight, a.loss_easy, scale, dy, dybj);
        if constexpr (C::MASKED) {   // last_X <- Y: the later segment's input gradient
          float lxf[D];
          to_full(std::integral_constant<int, M::Q0>{}, std::integral_constant<int, H>{},
                  std::integral_constant<int, D>{}, lx, lxf);
#pragma unroll
          for (int q = 0; q < DO; ++q) dy[q] += lxf[q];
        }
        if (g == 0) store_vec(has ? a.g_y + (size_t)r * DO : trash, dy);
        dec_adj(hn, dy, (uint32_t)k, NET_DEC, dh);
#pragma unroll
        for (int q = 0; q < M::QH; ++q) lam_hn[q] = lam_h[q] + dh[q];
        store_units<M::QH, H>(has ? a.g_hnew + (size_t)r * H : trash, lam_hn, g, trash);
        if constexpr (C::MASKED) {
          // h_new = encoder(x_in, M), x_in = X M + (1 - M) y_bj
          float xin[D], b0e[ES::Q0], a1[ES::Q1], a2[ES::Q1], dxq[MTID * 4], dxf[D];
          f32x4 din[MTID];
#pragma unroll
          for (int q = 0; q < D; ++q) xin[q] = x[q] * m[q] + (1.0f - m[q]) * ybj[q];
          fill_units<0, ES::Q0>(b0e, g, [&](auto U) {
            constexpr int u = decltype(U)::value;
            if constexpr (u < D) return tanh_f(xin[u]);
            else if constexpr (u < 2 * D) return m[u - D];
            else if constexpr (u == C::ENC_IN) return 1.0f;
            else return 0.0f;
          });
          uint32_t k1, k2;
          row_keep_bits<DROP>(a, gid, (uint32_t)k, NET_ENC, g, 16, k1, k2);
          mnet_hidden<ES, C::ACT, DROP>(Fe, b0e, a1, a2, k1, k2, a.dc.inv_keep, g);
          mnet_adj<ES, C::ACT, DROP, MTID>(Fe, lam_hn, a1, a2, k1, k2, a.dc.inv_keep, a.keep, din);
#pragma unroll
          for (int q = 0; q < MTID * 4; ++q) {
            const int u = 4 * q + g;
            const float th = q < ES::Q0 ? b0e[q < ES::Q0 ? q : 0] : 0.0f;
            float v = din[q / 4][q % 4] * (1.0f - th * th);
            if constexpr (C::ENC_CASE == 1) v += q < M::QH ? lam_hn[q < M::QH ? q : 0] : 0.0f;
            dxq[q] = u < D ? v : 0.0f;
          }
          to_full(std::integral_constant<int, MTID * 4>{}, std::integral_constant<int, 0>{},
                  std::integral_constant<int, D>{}, dxq, dxf);
#pragma unroll
          for (int q = 0; q < D; ++q) dybj[q] += dxf[q] * (1.0f - m[q]);
        }
        if (g == 0) store_vec(has ? a.g_ybj + (size_t)r * DO : trash, dybj);
        dec_adj(hp, dybj, (uint32_t)k, NET_DEC_BJ, lam_new);
        // commit for the chains that have this observation
#pragma unroll
        for (int q = 0; q < M::QH; ++q) lam_h[q] = has ? lam_new[q] : lam_h[q];
#pragma unroll
        for (int q = 0; q < M::Q0; ++q) lx[q] = has ? 0.0f : lx[q];
        const int nsrc = a.item_prev[r];
        if (has) {   // lane-local reload of the new segment's source
          src = nsrc;
          src_i = src >= 0 ? a.t_of_row[src >= 0 ? src : 0] : -1;
          load_source(src);
        }
      }
      --i;
    }
  }
  store_units<M::QH, H>(valid ? a.g_hstart + (size_t)b * H : trash, lam_h, g, trash);
}


// =====================================================================================
// Pass 2 of the lockstep backward on the matrix cores: weight gradients of the three
// networks from the adjoints the sweep stored.  Embarrassingly parallel over (step, path)
// pairs / observation rows, 16 per wave tile; 256-thread blocks with the fragments in LDS
// and one pair of dW staging images per wave.  The images are sized per network (the
// 41-dimensional shapes have up to 94 input units).
// =====================================================================================
template <class S> struct ImgRows {
  static constexpr int m1 = S::NT1 > S::NT0 ? S::NT1 : S::NT0;
  static constexpr int m2 = S::MT1 > S::MTO ? S::MT1 : S::MTO;
  static constexpr int ROWS = 16 * (m1 > m2 ? m1 : m2);
  static constexpr int FLOATS = ROWS * IMG_STRIDE;
};

// ODE network: one item per (step k, path b)
template <class C, bool DROP>
__global__ void __launch_bounds__(256, 1) k_ode_dw_pairs_mfma(KArgs a) {
  using M = MF<C>;
  using NL = typename C::Ode;
  using FR = OdeLdsFrags<C>;
  constexpr int NT1 = (M::W + 1 + 15) / 16, NT0 = (M::IN0 + 1 + 15) / 16;
  constexpr int RT = NT0 > NT1 ? (NT0 > M::MT1 ? NT0 : M::MT1) : (NT1 > M::MT1 ? NT1 : M::MT1);
  constexpr int IMG = 16 * RT * IMG_STRIDE;
  __shared__ __attribute__((aligned(16))) float lds_raw[4 * 2 * IMG + FR::NVEC * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int wave = blockIdx.x * 4 + wv, n_waves = gridDim.x * 4;
  lfp img_d = (lfp)lds_raw + wv * 2 * IMG, img_a = img_d + IMG;
  lfp fimg = (lfp)lds_raw + 4 * 2 * IMG;
  // (stored deltas, below: no product with a weight is left in this kernel -- no fragments to stage)
  const float* const drec = (a.chain || a.seg_chain) ? a.cdelta : nullptr;
  if (!drec) FR::stage(fimg, a.frag, threadIdx.x, 256);
  for (int i = threadIdx.x; i < 4 * 2 * IMG; i += 256) lds_raw[i] = 0.0f;
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
  const long long n_pairs = (long long)a.K * a.B;
  const long long n_tiles = (n_pairs + 15) / 16;
  // stored hidden activations of every pair, when the saving forward kept them (wave-uniform)
  const float* const rec = a.chain ? a.lact : (a.seg_chain ? a.act : nullptr);
  for (long long tile = wave; tile < n_tiles; tile += n_waves) {
    const long long p0 = tile * 16 + c;
    const bool valid = p0 < n_pairs;
    const long long p = valid ? p0 : 0;
    const int k = (int)(p / a.B), b = (int)(p % a.B);
    float h[M::QH], d3[M::QH], tx[C::D];
    load_units<M::QH, C::H>(a.ltraj + (size_t)p * C::H, h, g);
    load_units<M::QH, C::H>(a.lam_traj + (size_t)p * C::H, d3, g);
    const int src = a.src_row[p];
    const int sv = src >= 0 ? src : 0;
    const float* xp = src >= 0 ? (C::MASKED ? a.y_row + (size_t)sv * C::DO : a.X + (size_t)sv * C::D)
                               : a.start_X + (size_t)b * C::D;
#pragma unroll
    for (int q = 0; q < C::D; ++q) tx[q] = tanh_f(xp[q]);
    const float tsrc = a.n_obs > 0 ? a.time_f32[a.t_of_row[sv]] : 0.0f;
    const float tau = src >= 0 ? tsrc : 0.0f;
    const float dt = valid ? a.step_dt[k] : 0.0f, t = a.step_t[k];
#pragma unroll
    for (int q = 0; q < M::QH; ++q) d3[q] *= dt;
    float b0[M::Q0], a1[M::Q1], a2[M::Q1];
    in0_fill<C, 0>(b0, h, tx, tau, t - tau, g);
    uint32_t k1 = 0, k2 = 0;
    if (rec) {
      // (round 6) the saving forward was a wave-per-path / wave-per-item kernel (njode_chain.h,
      // njode_chain_seg.h): it stored both hidden activations of every (step, path) pair, one unit per
      // lane in the DPP layout -- unit 4 q + g in lane 16 g + q, i.e. THIS lane's Q1 units are
      // consecutive floats.  No recomputation (140 of this tile's ~440 matrix instructions) and no
      // mask draws: a dropped unit is stored as -0.0f.
      static_assert(M::Q1 <= 16, "one lane group's units fit its 16 lanes");
      const float* r1 = rec + ((size_t)p * 2) * 64 + 16 * g;
      float v1[16], v2[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 x1 = *(const f32x4*)(r1 + 4 * j), x2 = *(const f32x4*)(r1 + 64 + 4 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v1[4 * j + e] = x1[e];
          v2[4 * j + e] = x2[e];
        }
      }
#pragma unroll
      for (int q = 0; q < M::Q1; ++q) {
        a1[q] = v1[q];
        a2[q] = v2[q];
        if constexpr (DROP) {
          k1 |= (uint32_t)(__float_as_uint(v1[q]) != 0x80000000u) << q;
          k2 |= (uint32_t)(__float_as_uint(v2[q]) != 0x80000000u) << q;
        }
      }
      constexpr int QB = M::W / 4, GB = M::W % 4;   // the bias unit of the next layer's input
      a1[QB] = g == GB ? 1.0f : a1[QB];
      a2[QB] = g == GB ? 1.0f : a2[QB];
    } else {
    if constexpr (DROP) {
      const unsigned long long gid = a.gid0 + b;
      uint32_t st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1),
                               (uint32_t)k, NET_ODE);
      k1 = keep_bits<M::Q1>(st, a.dc.thr16);
      k2 = keep_bits<M::Q1>(st, a.dc.thr16);
    }
    F.begin();
    f32x4 acc[M::MT1];
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
    }
    F.begin();
    f32x4 acc[M::MT1];
    // layer 3
    img_write<M::QH>(img_d, d3, g, c);
    img_write<M::Q1>(img_a, a2, g, c);
    wave_lds_sync();
    dw_accumulate<M::MTH, NT1>(img_d, img_a, G3, g, c);
    float d2[M::QW], d1[M::QW];
    if (drec) {
      // (round 6) the wave-per-chain sweep stored delta2 and delta1 of every pair beside its adjoint,
      // in the activations' layout: this kernel is three outer-product accumulations and nothing else
      // (no transposed product: 96 of the remaining ~300 matrix instructions, and no fragment table)
      static_assert(M::QW <= 16, "one lane group's units fit its 16 lanes");
      const float* r2 = drec + ((size_t)p * 2) * 64 + 16 * g;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 x1 = *(const f32x4*)(r2 + 4 * j), x2 = *(const f32x4*)(r2 + 64 + 4 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (4 * j + e < M::QW) {
            d1[4 * j + e < M::QW ? 4 * j + e : 0] = valid ? x1[e] : 0.0f;
            d2[4 * j + e < M::QW ? 4 * j + e : 0] = valid ? x2[e] : 0.0f;
          }
        }
      }
      wave_lds_sync();
      // layer 2
      img_write<M::QW>(img_d, d2, g, c);
      img_write<M::Q1>(img_a, a1, g, c);
      wave_lds_sync();
      dw_accumulate<M::MT1, NT1>(img_d, img_a, G2, g, c);
      wave_lds_sync();
    } else {
#pragma unroll
    for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = zero4;
#pragma unroll
    for (int q = 0; q < M::QH; ++q)
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(F.b3(mt, q), d3[q], acc[mt]);
    hidden_delta<C, DROP>(acc, a2, d2, k2, a.dc.inv_keep, a.keep);
    wave_lds_sync();
    // layer 2
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
    hidden_delta<C, DROP>(acc, a1, d1, k1, a.dc.inv_keep, a.keep);
    wave_lds_sync();
    }
    // layer 1
    img_write<M::QW>(img_d, d1, g, c);
    img_write<M::Q0>(img_a, b0, g, c);
    wave_lds_sync();
    dw_accumulate<M::MT1, NT0>(img_d, img_a, G1, g, c);
    wave_lds_sync();
  }
  // flush in the parameter layout (in0 units back to the reference's column order)
  float* slab = a.slab + (size_t)wave * C::P + C::OFF_ODE;
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

// readout: two evaluations per observation row (after / before the jump)
template <class C, bool DROP>
__global__ void __launch_bounds__(256, 1) k_dec_dw_rows_mfma(KArgs a) {
  using S = typename DecS<C>::type;
  using NL = typename C::Dec;
  constexpr int IMG = ImgRows<S>::FLOATS;
  __shared__ __attribute__((aligned(16))) float lds_raw[4 * 2 * IMG + S::NALL * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int wave = blockIdx.x * 4 + wv, n_waves = gridDim.x * 4;
  lfp img_d = (lfp)lds_raw + wv * 2 * IMG, img_a = img_d + IMG;
  lfp fimg = (lfp)lds_raw + 4 * 2 * IMG;
  LdsFrags<S>::stage(fimg, a.frag_dec, threadIdx.x, 256);
  for (int i = threadIdx.x; i < 4 * 2 * IMG; i += 256) lds_raw[i] = 0.0f;
  __syncthreads();
  LdsFrags<S> F;
  F.init(fimg, lane);
  GradTiles<S> G;
  G.zero();
  const int n_items = 2 * a.n_obs;
  const int n_tiles = (n_items + 15) / 16;
  for (int tile = wave; tile < n_tiles; tile += n_waves) {
    const int it0 = tile * 16 + c;
    const bool valid = it0 < n_items;
    const int it = valid ? it0 : 0;
    const bool after = it < a.n_obs;
    const int r = after ? it : it - a.n_obs;
    const int b = a.obs_idx[r];
    const int kj = a.k_jump[a.t_of_row[r]];
    const float* hp = after ? a.ltraj + ((size_t)kj * a.B + b) * C::H : a.h_end + (size_t)r * C::H;
    const float* gp = after ? a.g_y + (size_t)r * C::DO : a.g_ybj + (size_t)r * C::DO;
    float b0[S::Q0], a1[S::Q1], a2[S::Q1], dq[S::QO];
    f32x4 out[S::MTO], din[1];
    dec_input<C, S>(hp, b0, g);
    load_units<S::QO, C::DO>(gp, dq, g);
#pragma unroll
    for (int q = 0; q < S::QO; ++q) dq[q] = valid ? dq[q] : 0.0f;
    uint32_t k1, k2;
    row_keep_bits<DROP>(a, a.gid0 + b, (uint32_t)kj, after ? NET_DEC : NET_DEC_BJ, g, S::Q1, k1, k2);
    mnet_fwd<S, C::ACT, DROP>(F, b0, a1, a2, out, k1, k2, a.dc.inv_keep, g);
    mnet_bwd<S, C::ACT, DROP, false, LdsFrags<S>, ImgRows<S>::ROWS>(
        F, G, img_d, img_a, dq, b0, a1, a2, k1, k2, a.dc.inv_keep, a.keep, din, g, c);
  }
  G.template flush<NL>(a.slab + (size_t)wave * C::P + C::OFF_DEC, g, c);
}

// encoder: every observation row (jump) and every start value
template <class C, bool DROP>
__global__ void __launch_bounds__(256, 1) k_enc_dw_rows_mfma(KArgs a) {
  using S = typename EncS<C>::type;
  using NL = typename C::Enc;
  constexpr int IMG = ImgRows<S>::FLOATS;
  __shared__ __attribute__((aligned(16))) float lds_raw[4 * 2 * IMG + S::NALL * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int wave = blockIdx.x * 4 + wv, n_waves = gridDim.x * 4;
  lfp img_d = (lfp)lds_raw + wv * 2 * IMG, img_a = img_d + IMG;
  lfp fimg = (lfp)lds_raw + 4 * 2 * IMG;
  LdsFrags<S>::stage(fimg, a.frag_enc, threadIdx.x, 256);
  for (int i = threadIdx.x; i < 4 * 2 * IMG; i += 256) lds_raw[i] = 0.0f;
  __syncthreads();
  LdsFrags<S> F;
  F.init(fimg, lane);
  GradTiles<S> G;
  G.zero();
  const int total = a.n_obs + a.B;
  const int n_tiles = (total + 15) / 16;
  for (int tile = wave; tile < n_tiles; tile += n_waves) {
    const int t0 = tile * 16 + c;
    const bool valid = t0 < total;
    const int tid = valid ? t0 : 0;
    const bool is_row = tid < a.n_obs;
    const int r = is_row ? tid : 0;
    const int b = is_row ? a.obs_idx[r] : tid - a.n_obs;
    const float* xp = is_row ? a.X + (size_t)r * C::D : a.start_X + (size_t)b * C::D;
    float b0[S::Q0], a1[S::Q1], a2[S::Q1], gh[S::QO];
#pragma unroll
    for (int q = 0; q < S::Q0; ++q) {
      const int u = 4 * q + g;
      float v = 0.0f;
      if (u < C::D) {
        float x = xp[u];
        if constexpr (C::MASKED) {
          const float mk = is_row ? a.M[(size_t)r * C::D + u] : 0.0f;
          const float yb = is_row ? a.ybj_row[(size_t)r * C::DO + (u < C::DO ? u : 0)] : 0.0f;
          x = is_row ? x * mk + (1.0f - mk) * yb : x;
        }
        v = tanh_f(x);
      } else if (C::MASKED && u < 2 * C::D) {
        v = is_row ? a.M[(size_t)r * C::D + (u - C::D)] : 0.0f;
      } else if (u == C::ENC_IN) {
        v = 1.0f;
      }
      b0[q] = v;
    }
    const float* gp = is_row ? a.g_hnew + (size_t)r * C::H : a.g_hstart + (size_t)b * C::H;
    load_units<S::QO, C::H>(gp, gh, g);
#pragma unroll
    for (int q = 0; q < S::QO; ++q) gh[q] = valid ? gh[q] : 0.0f;
    uint32_t k1, k2;
    row_keep_bits<DROP>(a, a.gid0 + b, is_row ? (uint32_t)a.k_jump[a.t_of_row[r]] : TKEY_START,
                        NET_ENC, g, S::Q1, k1, k2);
    f32x4 out[S::MTO], din[1];
    mnet_fwd<S, C::ACT, DROP>(F, b0, a1, a2, out, k1, k2, a.dc.inv_keep, g);
    mnet_bwd<S, C::ACT, DROP, false, LdsFrags<S>, ImgRows<S>::ROWS>(
        F, G, img_d, img_a, gh, b0, a1, a2, k1, k2, a.dc.inv_keep, a.keep, din, g, c);
  }
  G.template flush<NL>(a.slab + (size_t)wave * C::P + C::OFF_ENC, g, c);
}

}  // namespace njode
