// njode_lockstep_bwd.h -- exact backward of the lockstep plan (masked models, and any
// call the segment plan does not cover).
//
// In masked mode the state after a jump depends on the state before it
// (self-imputation, models.py:465-467) and the prediction becomes the next ODE input
// (models.py:483-484), so the adjoint is a genuinely sequential reverse sweep per path.
// It is split in two so that no kernel has to hold all three networks' gradient tiles:
//   pass 1  k_paths_bwd_adj : one lane per path, events in reverse; propagates the
//           adjoints through W^T products only and stores, per event, the gradient
//           at each network evaluation's output;
//   pass 2  k_ode_dw_pairs / k_dec_dw_rows / k_enc_dw_rows : embarrassingly parallel
//           weight-gradient kernels over (path, step) pairs / rows (recompute the
//           evaluation, outer products through the LDS tiles of njode_device.h).
#pragma once
#include "njode_kernels.h"

namespace njode {

// d/d x_in of the encoder from din (w.r.t. tanh(x) inputs) and the identity path
template <class C>
NJ_DEV void enc_input_grad(const float (&din)[C::D], const float (&ein)[C::ENC_IN],
                           const float (&gh)[C::H], float (&dx)[C::D]) {
#pragma unroll
  for (int q = 0; q < C::D; ++q) dx[q] = din[q] * (1.0f - ein[q] * ein[q]);
  if constexpr (C::ENC_CASE == 1) {
#pragma unroll
    for (int j = 0; j < C::H; ++j) dx[j % C::D] += gh[j];
  } else if constexpr (C::ENC_CASE == 2) {
    constexpr int mult = C::D / C::H;
#pragma unroll
    for (int q = 0; q < C::D; ++q) dx[q] += gh[q % C::H] * (1.0f / mult);
  }
}

// Backward of the GRU jump.  dhn = adjoint of the new state.  Produces the gate gradients
// (dgi, dgh: the deltas of the two affine maps) and the adjoint of the old state dh.
template <class C>
NJ_DEV void gru_gate_grads(const float (&dhn)[C::H], const float (&th)[C::H],
                           const float (&r)[C::H], const float (&z)[C::H], const float (&n)[C::H],
                           const float (&ghn)[C::H], float (&dgi)[3 * C::H],
                           float (&dgh)[3 * C::H], float (&dth)[C::H]) {
  constexpr int H = C::H;
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const float dn = dhn[i] * (1.0f - z[i]);
    const float dz = dhn[i] * (th[i] - n[i]);
    dth[i] = dhn[i] * z[i];
    const float dan = dn * (1.0f - n[i] * n[i]);      // pre-activation of n
    const float dr = dan * ghn[i];
    const float dar = dr * r[i] * (1.0f - r[i]);
    const float daz = dz * z[i] * (1.0f - z[i]);
    dgi[i] = dar;          dgh[i] = dar;
    dgi[H + i] = daz;      dgh[H + i] = daz;
    dgi[2 * H + i] = dan;  dgh[2 * H + i] = dan * r[i];
  }
}

// ---- pass 1: adjoint sweep ------------------------------------------------------------------
template <class C, bool DROP>
__global__ void __launch_bounds__(64, 1) k_paths_bwd_adj(KArgs a) {
  const int b0 = blockIdx.x * 64 + threadIdx.x;
  const bool valid = b0 < a.B;
  const int b = valid ? b0 : a.B - 1;
  const unsigned long long gid = a.gid0 + b;
  const cfp Po0 = as_cfp(a.P) + C::OFF_ODE, Pe0 = as_cfp(a.P) + C::OFF_ENC,
            Pd0 = as_cfp(a.P) + C::OFF_DEC;
  const cfp PTo0 = as_cfp(a.PT) + C::OFF_ODE, PTe0 = as_cfp(a.PT) + C::OFF_ENC,
            PTd0 = as_cfp(a.PT) + C::OFF_DEC;
  const int __attribute__((address_space(4)))* kjump =
      (const int __attribute__((address_space(4)))*)(unsigned long long)a.k_jump;
  const cfp sdt = as_cfp(a.step_dt), stt = as_cfp(a.step_t);
  float* const trash = a.trash + threadIdx.x * (C::H > C::D ? C::H : C::D);

  float lam_h[C::H], lam_x[C::D];
  // (the adjoint of the final state: the upstream gradient of hT, if the caller has one)
#pragma unroll
  for (int q = 0; q < C::H; ++q) lam_h[q] = (a.g_hT && valid) ? a.g_hT[(size_t)b * C::H + q] : 0.0f;
#pragma unroll
  for (int q = 0; q < C::D; ++q) lam_x[q] = 0.0f;
  int src = a.last_row[b];
  int src_i = (src >= 0) ? a.t_of_row[src] : -1;
  int i = a.n_times - 1;
  float a1[C::W], a2[C::W];
  Masks<C, DROP> mk;

  for (int k = a.K; k >= 0; --k) {
    if (k < a.K) {
      // ---- reverse Euler step k
      float h[C::H], tx[C::D], in0[C::ODE_IN], f[C::H], dout[C::H], din[C::D + C::H];
      load_vec(a.ltraj + ((size_t)k * a.B + b) * C::H, h);
      const int sv = src >= 0 ? src : 0;
      const float* xp = src >= 0 ? (C::MASKED ? a.y_row + (size_t)sv * C::DO : a.X + (size_t)sv * C::D)
                                 : a.start_X + (size_t)b * C::D;
      const float tsrc = a.n_obs > 0 ? a.time_f32[a.t_of_row[sv]] : 0.0f;
      const float tau = src >= 0 ? tsrc : 0.0f;
#pragma unroll
      for (int q = 0; q < C::D; ++q) tx[q] = tanh_f(xp[q]);
      const float dt = sdt[k], t = stt[k];
      ode_input<C>(tx, h, tau, t, in0);
      mk.draw(a, gid, (uint32_t)k, NET_ODE);
      net_fwd<typename C::Ode, C::ACT, DROP>(Po0, in0, f, a1, a2, mk.m1, mk.m2, a.dc.inv_keep);
      store_vec(valid ? a.lam_traj + ((size_t)k * a.B + b) * C::H : trash, lam_h);
#pragma unroll
      for (int q = 0; q < C::H; ++q) dout[q] = dt * lam_h[q];
      net_bwd_inputs<typename C::Ode, C::ACT, DROP, 0, C::D + C::H>(
          PTo0, dout, a1, a2, mk.m1, mk.m2, a.dc.inv_keep, a.keep, din);
#pragma unroll
      for (int q = 0; q < C::H; ++q) {
        const float th = in0[C::D + q];
        lam_h[q] = fmaf(din[C::D + q], 1.0f - th * th, lam_h[q]);
      }
#pragma unroll
      for (int q = 0; q < C::D; ++q) lam_x[q] = fmaf(din[q], 1.0f - tx[q] * tx[q], lam_x[q]);
    }
    // ---- reverse the jump the forward pass applied right before step k
    while (i >= 0 && kjump[i] == k) {
      const bool has = valid && src >= 0 && src_i == i;
      if (__any(has)) {
        if (has) {
          const int r = src;
          float hn[C::H], hp[C::H], th[C::H], x[C::D], mask[C::D], y[C::DO], ybj[C::DO];
          float dy[C::DO], dybj[C::DO], dinh[C::H], dh[C::H], lam_hn[C::H];
          load_vec(a.ltraj + ((size_t)k * a.B + b) * C::H, hn);   // state after the jump
          load_vec(a.h_end + (size_t)r * C::H, hp);               // state before the jump
          load_vec(a.X + (size_t)r * C::D, x);
          if constexpr (C::MASKED) {
            load_vec(a.M + (size_t)r * C::D, mask);
          } else {
#pragma unroll
            for (int q = 0; q < C::D; ++q) mask[q] = 1.0f;
          }
          load_vec(a.y_row + (size_t)r * C::DO, y);
          load_vec(a.ybj_row + (size_t)r * C::DO, ybj);
          const float scale = a.inv_batch * __builtin_amdgcn_rcpf((float)a.n_obs_ot[b]);
          loss_row<C>(x, mask, y, ybj, a.weight, a.loss_easy, scale, dy, dybj);
          if constexpr (C::MASKED) {   // last_X <- Y: the next segment's input gradient
#pragma unroll
            for (int q = 0; q < C::DO; ++q) dy[q] += lam_x[q];
          }
          store_vec(a.g_y + (size_t)r * C::DO, dy);
          // y = readout(h_new)
          {
            float yy[C::DO];
            mk.draw(a, gid, (uint32_t)k, NET_DEC);
            readout<C, DROP>(Pd0, hn, th, a1, a2, mk, a.dc.inv_keep, yy);
            net_bwd_inputs<typename C::Dec, C::ACT, DROP, 0, C::H>(
                PTd0, dy, a1, a2, mk.m1, mk.m2, a.dc.inv_keep, a.keep, dinh);
            readout_input_grad<C>(dinh, th, dy, dh);
          }
#pragma unroll
          for (int q = 0; q < C::H; ++q) lam_hn[q] = lam_h[q] + dh[q];
          store_vec(a.g_hnew + (size_t)r * C::H, lam_hn);
          // h_new = encoder(x_in, M), x_in = X M + (1 - M) y_bj   (masked only)
          if constexpr (C::MASKED) {
            float xin[C::D], ein[C::ENC_IN], hh[C::H], dinx[C::D], dx[C::D];
#pragma unroll
            for (int q = 0; q < C::D; ++q) xin[q] = x[q] * mask[q] + (1.0f - mask[q]) * ybj[q];
            mk.draw(a, gid, (uint32_t)k, NET_ENC);
            encode<C, DROP>(Pe0, xin, mask, ein, a1, a2, mk, a.dc.inv_keep, hh);
            net_bwd_inputs<typename C::Enc, C::ACT, DROP, 0, C::D>(
                PTe0, lam_hn, a1, a2, mk.m1, mk.m2, a.dc.inv_keep, a.keep, dinx);
            enc_input_grad<C>(dinx, ein, lam_hn, dx);
#pragma unroll
            for (int q = 0; q < C::D; ++q) dybj[q] += dx[q] * (1.0f - mask[q]);
          }
          store_vec(a.g_ybj + (size_t)r * C::DO, dybj);
          // y_bj = readout(h_pre): without the GRU the only path from the state before
          // the jump
          {
            float yy[C::DO];
            mk.draw(a, gid, (uint32_t)k, NET_DEC_BJ);
            readout<C, DROP>(Pd0, hp, th, a1, a2, mk, a.dc.inv_keep, yy);
            net_bwd_inputs<typename C::Dec, C::ACT, DROP, 0, C::H>(
                PTd0, dybj, a1, a2, mk.m1, mk.m2, a.dc.inv_keep, a.keep, dinh);
            readout_input_grad<C>(dinh, th, dybj, lam_h);
          }
          if constexpr (C::RNN) {
            // h_new = GRU(tanh(X), tanh(h_pre)): second path into the old state
            float gtx[C::D], gth[C::H], gr[C::H], gz[C::H], gn[C::H], gg[C::H], hh[C::H];
            float dgi[3 * C::H], dgh[3 * C::H], dth[C::H], dthw[C::H];
            gru_jump<C>(as_cfp(a.P) + C::OFF_GRU, x, hp, gtx, gth, gr, gz, gn, gg, hh);
            gru_gate_grads<C>(lam_hn, gth, gr, gz, gn, gg, dgi, dgh, dth);
            dense_T_range<C::H, 3 * C::H, 0, C::H>(
                launder(as_cfp(a.PT) + C::OFF_GRU + C::G_WHH), dgh, dthw);
#pragma unroll
            for (int q = 0; q < C::H; ++q)
              lam_h[q] += (dth[q] + dthw[q]) * (1.0f - gth[q] * gth[q]);
          }
#pragma unroll
          for (int q = 0; q < C::D; ++q) lam_x[q] = 0.0f;
          src = a.item_prev[r];
          src_i = (src >= 0) ? a.t_of_row[src >= 0 ? src : 0] : -1;
        }
      }
      --i;
    }
  }
  store_vec(valid ? a.g_hstart + (size_t)b * C::H : trash, lam_h);
}

// ---- pass 2: weight gradients -----------------------------------------------------------------
// ODE network: one item per (step k, path b)
template <class C, bool DROP>
__global__ void __launch_bounds__(64, 1) k_ode_dw_pairs(KArgs a) {
  using NL = typename C::Ode;
  __shared__ __attribute__((aligned(16))) float lds_raw[NetAcc<NL>::LDS_FLOATS];
  lfp lds = (lfp)lds_raw;
  const int lane = threadIdx.x, wave = blockIdx.x;
  NetAcc<NL> g;
  g.zero();
  const cfp Po0 = as_cfp(a.P) + C::OFF_ODE, PTo0 = as_cfp(a.PT) + C::OFF_ODE;
  const long long n_pairs = (long long)a.K * a.B;
  const long long n_tiles = (n_pairs + 63) / 64;
  for (long long tile = wave; tile < n_tiles; tile += a.n_waves) {
    const long long p0 = tile * 64 + lane;
    const bool valid = p0 < n_pairs;
    const long long p = valid ? p0 : 0;
    const int k = (int)(p / a.B), b = (int)(p % a.B);
    float h[C::H], tx[C::D], in0[C::ODE_IN], a1[C::W], a2[C::W], f[C::H], dout[C::H], din[1];
    load_vec(a.ltraj + (size_t)p * C::H, h);
    const int src = a.src_row[p];
    const int sv = src >= 0 ? src : 0;
    const float* xp = src >= 0 ? (C::MASKED ? a.y_row + (size_t)sv * C::DO : a.X + (size_t)sv * C::D)
                               : a.start_X + (size_t)b * C::D;
    const float tsrc = a.n_obs > 0 ? a.time_f32[a.t_of_row[sv]] : 0.0f;
    const float tau = src >= 0 ? tsrc : 0.0f;
#pragma unroll
    for (int q = 0; q < C::D; ++q) tx[q] = tanh_f(xp[q]);
    const float dt = valid ? a.step_dt[k] : 0.0f, t = a.step_t[k];
    ode_input<C>(tx, h, tau, t, in0);
    Masks<C, DROP> mk;
    mk.draw(a, a.gid0 + b, (uint32_t)k, NET_ODE);
    net_fwd<NL, C::ACT, DROP>(Po0, in0, f, a1, a2, mk.m1, mk.m2, a.dc.inv_keep);
#pragma unroll
    for (int q = 0; q < C::H; ++q) dout[q] = dt * a.lam_traj[(size_t)p * C::H + q];
    net_bwd<NL, C::ACT, DROP, 0, 0>(PTo0, lds, g, in0, dout, a1, a2, mk.m1, mk.m2, a.dc.inv_keep,
                                    a.keep, din, lane);
  }
  g.flush(a.slab + (size_t)wave * C::P + C::OFF_ODE, lane);
}

// readout: two evaluations per observation row (after / before the jump)
template <class C, bool DROP>
__global__ void __launch_bounds__(64, 1) k_dec_dw_rows(KArgs a) {
  using NL = typename C::Dec;
  __shared__ __attribute__((aligned(16))) float lds_raw[NetAcc<NL>::LDS_FLOATS];
  lfp lds = (lfp)lds_raw;
  const int lane = threadIdx.x, wave = blockIdx.x;
  NetAcc<NL> g;
  g.zero();
  const cfp Pd0 = as_cfp(a.P) + C::OFF_DEC, PTd0 = as_cfp(a.PT) + C::OFF_DEC;
  const int n_items = 2 * a.n_obs;
  const int n_tiles = (n_items + 63) / 64;
  for (int tile = wave; tile < n_tiles; tile += a.n_waves) {
    const int it0 = tile * 64 + lane;
    const bool valid = it0 < n_items;
    const int it = valid ? it0 : 0;
    const bool after = it < a.n_obs;
    const int r = after ? it : it - a.n_obs;
    const int b = a.obs_idx[r];
    const int kj = a.k_jump[a.t_of_row[r]];
    const float* hp = after ? a.ltraj + ((size_t)kj * a.B + b) * C::H : a.h_end + (size_t)r * C::H;
    const float* gp = after ? a.g_y + (size_t)r * C::DO : a.g_ybj + (size_t)r * C::DO;
    float h[C::H], th[C::H], a1[C::W], a2[C::W], y[C::DO], dout[C::DO], din[1];
    load_vec(hp, h);
#pragma unroll
    for (int q = 0; q < C::DO; ++q) dout[q] = valid ? gp[q] : 0.0f;
    Masks<C, DROP> mk;
    mk.draw(a, a.gid0 + b, (uint32_t)kj, after ? NET_DEC : NET_DEC_BJ);
    readout<C, DROP>(Pd0, h, th, a1, a2, mk, a.dc.inv_keep, y);
    net_bwd<NL, C::ACT, DROP, 0, 0>(PTd0, lds, g, th, dout, a1, a2, mk.m1, mk.m2, a.dc.inv_keep,
                                    a.keep, din, lane);
  }
  g.flush(a.slab + (size_t)wave * C::P + C::OFF_DEC, lane);
}

// encoder: every observation row (jump) and every start value
template <class C, bool DROP>
__global__ void __launch_bounds__(64, 1) k_enc_dw_rows(KArgs a) {
  using NL = typename C::Enc;
  __shared__ __attribute__((aligned(16))) float lds_raw[NetAcc<NL>::LDS_FLOATS];
  lfp lds = (lfp)lds_raw;
  const int lane = threadIdx.x, wave = blockIdx.x;
  NetAcc<NL> g;
  g.zero();
  const cfp Pe0 = as_cfp(a.P) + C::OFF_ENC, PTe0 = as_cfp(a.PT) + C::OFF_ENC;
  // with the GRU jump the encoder only produces the start states
  const int first = C::RNN ? a.n_obs : 0;
  const int total = a.n_obs + a.B;
  const int n_tiles = (total - first + 63) / 64;
  for (int tile = wave; tile < n_tiles; tile += a.n_waves) {
    const int t0 = first + tile * 64 + lane;
    const bool valid = t0 < total;
    const int tid = valid ? t0 : first;
    const bool is_row = tid < a.n_obs;
    const int r = is_row ? tid : 0;
    const int b = is_row ? a.obs_idx[r] : tid - a.n_obs;
    float x[C::D], mask[C::D], ein[C::ENC_IN], a1[C::W], a2[C::W], h[C::H], gh[C::H], din[1];
#pragma unroll
    for (int q = 0; q < C::D; ++q) mask[q] = 0.0f;
    if (is_row) {
      load_vec(a.X + (size_t)r * C::D, x);
      if constexpr (C::MASKED) {
        load_vec(a.M + (size_t)r * C::D, mask);
#pragma unroll
        for (int q = 0; q < C::D; ++q)
          x[q] = x[q] * mask[q] + (1.0f - mask[q]) * a.ybj_row[(size_t)r * C::DO + q];
      }
    } else {
      load_vec(a.start_X + (size_t)b * C::D, x);
    }
    const float* gp = is_row ? a.g_hnew + (size_t)r * C::H : a.g_hstart + (size_t)b * C::H;
#pragma unroll
    for (int q = 0; q < C::H; ++q) gh[q] = valid ? gp[q] : 0.0f;
    Masks<C, DROP> mk;
    mk.draw(a, a.gid0 + b, is_row ? (uint32_t)a.k_jump[a.t_of_row[r]] : TKEY_START, NET_ENC);
    encode<C, DROP>(Pe0, x, mask, ein, a1, a2, mk, a.dc.inv_keep, h);
    net_bwd<NL, C::ACT, DROP, 0, 0>(PTe0, lds, g, ein, gh, a1, a2, mk.m1, mk.m2, a.dc.inv_keep,
                                    a.keep, din, lane);
  }
  g.flush(a.slab + (size_t)wave * C::P + C::OFF_ENC, lane);
}

// GRU parameters: outer products of the gate gradients with tanh(X) / tanh(h_pre)
template <class C>
__global__ void __launch_bounds__(64, 1) k_gru_dw_rows(KArgs a) {
  using TI = Tile<3 * C::H, C::D>;
  using TH = Tile<3 * C::H, C::H>;
  constexpr int LDSF = TI::LDS_FLOATS > TH::LDS_FLOATS ? TI::LDS_FLOATS : TH::LDS_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds_raw[LDSF];
  lfp lds = (lfp)lds_raw;
  const int lane = threadIdx.x, wave = blockIdx.x;
  float acc_i[TI::NACC], acc_h[TH::NACC];
#pragma unroll
  for (int q = 0; q < TI::NACC; ++q) acc_i[q] = 0.0f;
#pragma unroll
  for (int q = 0; q < TH::NACC; ++q) acc_h[q] = 0.0f;
  const int n_tiles = (a.n_obs + 63) / 64;
  for (int tile = wave; tile < n_tiles; tile += a.n_waves) {
    const int r0 = tile * 64 + lane;
    const bool valid = r0 < a.n_obs;
    const int r = valid ? r0 : 0;
    float x[C::D], hp[C::H], lam[C::H], gtx[C::D], gth[C::H], gr[C::H], gz[C::H], gn[C::H],
        gg[C::H], hh[C::H], dgi[3 * C::H], dgh[3 * C::H], dth[C::H];
    load_vec(a.X + (size_t)r * C::D, x);
    load_vec(a.h_end + (size_t)r * C::H, hp);
#pragma unroll
    for (int q = 0; q < C::H; ++q) lam[q] = valid ? a.g_hnew[(size_t)r * C::H + q] : 0.0f;
    gru_jump<C>(as_cfp(a.P) + C::OFF_GRU, x, hp, gtx, gth, gr, gz, gn, gg, hh);
    gru_gate_grads<C>(lam, gth, gr, gz, gn, gg, dgi, dgh, dth);
    TI::update(lds, acc_i, dgi, gtx, lane);
    TH::update(lds, acc_h, dgh, gth, lane);
  }
  float* slab = a.slab + (size_t)wave * C::P + C::OFF_GRU;
  TI::flush(acc_i, slab + C::G_WIH, slab + C::G_BIH, lane);
  TH::flush(acc_h, slab + C::G_WHH, slab + C::G_BHH, lane);
}

}  // namespace njode
