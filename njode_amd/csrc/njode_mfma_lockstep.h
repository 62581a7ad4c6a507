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

}  // namespace njode
