// njode_chain_seg.h -- the segment plan's ODE kernels in the LATENCY regime: ONE WAVE PER ITEM.
//
// Why (round 6).  demo.py:81 trains at B = 100, the shipped models at 200: ~1 000 .. 2 000 items of
// ~9 Euler steps whose LONGEST one (~50 steps) sets the time of both ODE kernels -- 139 of the step's
// 201 us at B = 100 (profiles/r05_small_batch_kernels.txt), at 1.3 - 1.5 us per step of a 16-wide MFMA
// tile over four waves (two workgroup barriers and two LDS all-gathers per step).  Here an item is
// one wave, a lane is a unit, the weights are register resident and the layer inputs travel through
// the DPP modifier of the fma (njode_dpp.h) -- and for the demo models' small state (H <= 16):
//
//   * h and everything H-wide (tanh h, f, the adjoint, delta3) is kept ROW-REPLICATED: unit o in lane o
//     of each of the wave's four rows.  Layer 1 then reads its state inputs with row_newbcast on that
//     register (H fma, no replication), and the H-wide OUTPUT layer is a K-split: row g sums the
//     hidden units = g (mod 4) from its own slice of the unit-layout activation (13 fma instead of
//     50) and dpp_rows_sum adds the four partial sums into every row;
//   * only the W x W layer needs the four replicated registers (dpp_replicate, 50 fma);
//   * the x and tau columns of layer 1 are constant along an item: folded into one accumulator.
//
// ~75 fma + 3 tanh per Euler step forward, ~75 + 1 tanh in the sweep, nothing else on the chain.
// Weight gradients are NOT accumulated in the sweep (110 more fma per step on the critical path of the
// longest item): the sweep stores the adjoint after every step in the lockstep plan's (step, path)
// layout and the lockstep plan's parallel weight-gradient kernel (k_ode_dw_pairs_mfma,
// njode_mfma_lockstep.h) turns them into dW over the whole chip.  An item covers the pairs (k, b) of
// its steps; the pairs behind a path's last observation belong to no item and are filled with
// zeros (adjoint 0: no contribution) by one more wave per path -- which is also the wave that
// evolves the path's tail to hT when the caller wants it.
//
// Buffers: ltraj [K+1][B][H], lam_traj [K][B][H], src_row [K][B] (the lockstep plan's), records
// act [(k B + b)][layer][64 lanes], keep masks dbits [(b K + k)][layer] (lane masks, njode_chain.h),
// h_end / lam_end / lam_start per row (the segment plan's).
#pragma once
#include "njode_chain.h"

namespace njode {

// keep masks of the ODE network for every (path, step) (seg_chain_bits_body, njode_mfma.h): a launch of
// its own only when the fragment-pack launch did not carry them (KArgs::dbits_ready)
template <class C> __global__ void __launch_bounds__(256) k_seg_chain_bits(KArgs a) {
  seg_chain_bits_body<C>(a, blockIdx.x, gridDim.x);
}

// wave w < n_obs: the item of row w; wave n_obs + b: path b's tail
template <class C, bool DROP>
__device__ __forceinline__ void seg_fwd_chain_wave(const KArgs& a, int tails, int wave) {
  using NO = typename C::Ode;
  constexpr int D = C::D, H = C::H, W = C::W, OIN = C::ODE_IN, NQW = (W + 3) / 4;
  const int lane = threadIdx.x & 63, u = dpp_unit(lane), g = lane >> 4, c = lane & 15;
  if (wave >= a.n_obs + a.B) return;
  const bool is_tail = wave >= a.n_obs;
  const bool SAVE = a.save_traj != 0;
  const float* Po = a.P + C::OFF_ODE;
  typedef const int __attribute__((address_space(4)))* cip;
  const cip obs_idx = (cip)(unsigned long long)a.obs_idx,
            item_len = (cip)(unsigned long long)a.item_len, item_kbeg = (cip)(unsigned long long)a.item_kbeg,
            item_prev = (cip)(unsigned long long)a.item_prev, last_row = (cip)(unsigned long long)a.last_row,
            t_of_row = (cip)(unsigned long long)a.t_of_row, kjump = (cip)(unsigned long long)a.k_jump;
  const cfp tf32 = as_cfp(a.time_f32), sdt = as_cfp(a.step_dt), stt = as_cfp(a.step_t);

  // ---- the item (wave-uniform)
  int r, b, n, kbeg, prev;
  if (!is_tail) {
    r = wave;   // (every wave has a SIMD slot of its own: no need for the plan's order by length)
    b = obs_idx[r];
    n = item_len[r];
    kbeg = item_kbeg[r];
    prev = item_prev[r];
  } else {
    b = r = wave - a.n_obs;
    prev = last_row[b];
    kbeg = prev >= 0 ? kjump[t_of_row[prev]] : 0;
    n = a.K - kbeg;
  }
  const float tau = prev >= 0 ? tf32[t_of_row[prev]] : 0.0f;
  const float* xp = prev >= 0 ? a.X + (size_t)prev * D : a.start_X + (size_t)b * D;
  const float* h0p = prev >= 0 ? a.h0row + (size_t)prev * H : a.h0start + (size_t)b * H;

  if (is_tail && SAVE) {
    // the pairs behind the path's last observation belong to no item: state 0, adjoint 0, source row -1
    // (what k_ode_dw_pairs_mfma reads of them must be finite; their contribution is 0 * finite)
    for (int i = lane; i < n * H; i += 64) {
      const int k = kbeg + i / H, q = i % H;
      a.ltraj[((size_t)k * a.B + b) * H + q] = 0.0f;
      a.lam_traj[((size_t)k * a.B + b) * H + q] = 0.0f;
    }
    for (int i = lane; i < n; i += 64) a.src_row[(size_t)(kbeg + i) * a.B + b] = -1;
    // ... and so must the stored activations be that the pair kernel reads back (0 * NaN is NaN)
    for (int s = 0; s < n; ++s) {
      float* rec = a.act + ((size_t)(kbeg + s) * a.B + b) * CHAIN_ACT_FLOATS;
      rec[lane] = 0.0f;
      rec[64 + lane] = 0.0f;
      if (a.cdelta) {
        float* dr = a.cdelta + ((size_t)(kbeg + s) * a.B + b) * CHAIN_ACT_FLOATS;
        dr[lane] = 0.0f;
        dr[64 + lane] = 0.0f;
      }
    }
  }
  if (is_tail && !tails) return;

  // ---- the lane's weights.  Unit layout (unit u = 4 c + g): row u of W1 (state columns) and W2;
  // K-split layout of the output layer: lane (g, c) holds W3[c][4 n + g], n = 0 .. NQW-1
  const bool uW = u < W, cH = c < H;
  const int jW = uW ? u : 0, jc = cH ? c : 0;
  float w1h[H], w2[W], w3p[NQW], w1x[D];
  chain_load_row<H>(w1h, Po + NO::woff(0) + (size_t)jW * OIN + D, uW, 1);
  chain_load_row<D>(w1x, Po + NO::woff(0) + (size_t)jW * OIN, uW, 1);
  chain_load_row<W>(w2, Po + NO::woff(1) + (size_t)jW * W, uW, 1);
#pragma unroll
  for (int q = 0; q < NQW; ++q) w3p[q] = (cH && 4 * q + g < W) ? Po[NO::woff(2) + (size_t)jc * W + 4 * q + g] : 0.0f;
  const float w1tau = uW ? Po[NO::woff(0) + (size_t)jW * OIN + D + H] : 0.0f;
  const float w1td = uW ? Po[NO::woff(0) + (size_t)jW * OIN + D + H + 1] : 0.0f;
  const float w1ct = (C::CURT && uW) ? Po[NO::woff(0) + (size_t)jW * OIN + D + H + 2] : 0.0f;
  const float ob1 = uW ? Po[NO::boff(0) + jW] : 0.0f, ob2 = uW ? Po[NO::boff(1) + jW] : 0.0f;
  const float ob3 = (cH && g == 0) ? Po[NO::boff(2) + jc] : 0.0f;   // (once: row 0's partial sum carries it)

  // the item's constant part of layer 1: b1 + W1x tanh(x) + w_tau tau
  float c1 = ob1;
#pragma unroll
  for (int i = 0; i < D; ++i) c1 = fmaf(w1x[i], tanh_f(xp[i]), c1);
  c1 = fmaf(w1tau, tau, c1);

  float h = cH ? h0p[jc] : 0.0f;   // row-replicated: unit c in lane c of every row
  float th = tanh_f(h);

  float* const trash = a.trash + threadIdx.x;
  const bool sv = SAVE && !is_tail;
  float* lt_p = (sv && cH && g == 0) ? a.ltraj + ((size_t)kbeg * a.B + b) * H + jc : trash;
  const size_t lt_step = (sv && cH && g == 0) ? (size_t)a.B * H : 0;
  float* la_p = sv ? a.act + ((size_t)kbeg * a.B + b) * CHAIN_ACT_FLOATS + lane : trash;
  const size_t la_step = sv ? (size_t)a.B * CHAIN_ACT_FLOATS : 0;
  const int la_2 = sv ? 64 : 0;
  int* sr_p = (sv && lane == 0) ? a.src_row + (size_t)kbeg * a.B + b : (int*)trash;
  const size_t sr_step = (sv && lane == 0) ? (size_t)a.B : 0;

  typedef const unsigned long long __attribute__((address_space(4)))* cu64p;
  const cu64p sbits = (cu64p)(unsigned long long)a.dbits;
  const float inv_keep = a.dc.inv_keep;
  float dt_n = n > 0 ? sdt[kbeg] : 0.0f, t_n = n > 0 ? stt[kbeg] : 0.0f;
  uint64_t m1_n = 0, m2_n = 0;
  if constexpr (DROP) {
    if (n > 0) {
      m1_n = sbits[chain_step_bits(kbeg, a.K, b)];
      m2_n = sbits[chain_step_bits(kbeg, a.K, b) + 1];
    }
  }
  for (int s = 0; s < n; ++s) {
    const int k = kbeg + s;
    *lt_p = h;   // state before step k
    lt_p += lt_step;
    *sr_p = prev;
    sr_p += sr_step;
    const float dt = dt_n, t = t_n;
    const uint64_t m1 = m1_n, m2 = m2_n;
    {
      const int kn = s + 1 < n ? k + 1 : k;
      dt_n = sdt[kn];
      t_n = stt[kn];
      if constexpr (DROP) {
        m1_n = sbits[chain_step_bits(kn, a.K, b)];
        m2_n = sbits[chain_step_bits(kn, a.K, b) + 1];
      }
    }
    // layer 1: the state inputs from the row-replicated tanh(h)
    float z = fmaf(w1td, t - tau, c1);
    if constexpr (C::CURT) z = fmaf(w1ct, tau + (t - tau), z);
    dpp_settle(th);
    z = dpp_row_dot<H>(z, th, w1h);
    const float a1l = chain_hidden<C::ACT, DROP>(z, m1, inv_keep);
    // layer 2: W x W on the four replicated registers
    float R[4];
    dpp_replicate(a1l, R);
    z = dpp_dot<W>(ob2, R, w2);
    float a2l = chain_hidden<C::ACT, DROP>(z, m2, inv_keep);
    la_p[0] = a1l;
    la_p[la_2] = a2l;
    la_p += la_step;
    // layer 3, K-split: this row's hidden units, then the four rows' partial sums
    dpp_settle(a2l);
    const float f = dpp_rows_sum(dpp_row_dot<NQW>(ob3, a2l, w3p));
    h = cH ? fmaf(dt, f, h) : 0.0f;
    th = tanh_f(h);
  }
  if (cH && g == 0) {
    float* out = is_tail ? a.hT + (size_t)b * H : a.h_end + (size_t)r * H;
    out[c] = h;
  }
}

template <class C, bool DROP>
__global__ void __launch_bounds__(256) k_seg_fwd_chain(KArgs a, int tails) {
  seg_fwd_chain_wave<C, DROP>(a, tails, (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}
// ... with the NEXT batch's plan in front of the launch's own blocks (njode_plan.h, NJODE_C_PLAN_DEFER)
template <class C, bool DROP>
__global__ void __launch_bounds__(256) k_seg_fwd_chain_plan(KArgs a, int tails, PlanJob job) {
  __shared__ int lds_plan[PLAN_LDS_INTS];
  if ((int)blockIdx.x < job.P) {
    plan_grid_body(job, blockIdx.x, lds_plan);
    return;
  }
  seg_fwd_chain_wave<C, DROP>(a, tails, ((int)blockIdx.x - job.P) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}

// reverse sweep of every item: lam_end[row] -> lam_start[row], the adjoint after every step ->
// lam_traj (the weight gradients: k_ode_dw_pairs_mfma)
template <class C, bool DROP>
__global__ void __launch_bounds__(256) k_seg_bwd_chain(KArgs a) {
  using NO = typename C::Ode;
  constexpr int D = C::D, H = C::H, W = C::W, OIN = C::ODE_IN, NQW = (W + 3) / 4;
  const int lane = threadIdx.x & 63, u = dpp_unit(lane), g = lane >> 4, c = lane & 15;
  const int wave = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= a.n_obs) return;
  const float* Po = a.P + C::OFF_ODE;
  typedef const int __attribute__((address_space(4)))* cip;
  const cip obs_idx = (cip)(unsigned long long)a.obs_idx,
            item_len = (cip)(unsigned long long)a.item_len, item_kbeg = (cip)(unsigned long long)a.item_kbeg;
  const cfp sdt = as_cfp(a.step_dt), stt = as_cfp(a.step_t), tf32 = as_cfp(a.time_f32);
  const int r = wave, b = obs_idx[r], n = item_len[r], kbeg = item_kbeg[r];
  // the item's sums S0 = sum delta1, S1 = sum delta1 (t - tau) for the x / tau / time columns of dW1
  // (njode_chain_dw.h); tau = the time of the observation the item starts from
  const cip item_prev = (cip)(unsigned long long)a.item_prev, t_of_row = (cip)(unsigned long long)a.t_of_row;
  const int prev = item_prev[r];
  const float tau_s = prev >= 0 ? tf32[t_of_row[prev >= 0 ? prev : 0]] : 0.0f;
  float s0acc = 0.0f, s1acc = 0.0f;

  // transposed rows: unit layout W3[o][u] (o < H), W2[i][u]; K-split of W1^T: lane (g, c) holds
  // W1[4 n + g][D + c] (the state column c)
  const bool uW = u < W, cH = c < H;
  const int jW = uW ? u : 0, jc = cH ? c : 0;
  float w3t[H], w2t[W], w1tp[NQW];
  chain_load_row<H>(w3t, Po + NO::woff(2) + jW, uW, W);
  chain_load_row<W>(w2t, Po + NO::woff(1) + jW, uW, W);
#pragma unroll
  for (int q = 0; q < NQW; ++q)
    w1tp[q] = (cH && 4 * q + g < W) ? Po[NO::woff(0) + (size_t)(4 * q + g) * OIN + D + jc] : 0.0f;
  const float inv_keep = a.dc.inv_keep, keepf = a.keep;

  float lam = cH ? a.lam_end[(size_t)r * H + jc] : 0.0f;   // row-replicated
  float* const trash = a.trash + threadIdx.x;
  const int klast = kbeg + (n > 0 ? n - 1 : 0);
  const float* lt_p = a.ltraj + ((size_t)klast * a.B + b) * H + jc;
  const float* const lt_0 = a.ltraj + ((size_t)kbeg * a.B + b) * H + jc;
  float* lm_p = (cH && g == 0) ? a.lam_traj + ((size_t)klast * a.B + b) * H + jc : trash;
  const size_t lt_back = (size_t)a.B * H, lm_back = (cH && g == 0) ? lt_back : 0;
  float* ld_p = a.cdelta ? a.cdelta + ((size_t)klast * a.B + b) * CHAIN_ACT_FLOATS + lane : trash;
  const size_t ld_back = a.cdelta ? (size_t)a.B * CHAIN_ACT_FLOATS : 0;
  const float* la_p = a.act + ((size_t)klast * a.B + b) * CHAIN_ACT_FLOATS + lane;
  const float* const la_0 = a.act + ((size_t)kbeg * a.B + b) * CHAIN_ACT_FLOATS + lane;
  const size_t la_back = (size_t)a.B * CHAIN_ACT_FLOATS;
  // two steps ahead, two register sets with compile-time indices (njode_chain.h, the sweep)
  float hb[2], a1b[2], a2b[2];
  auto fetch = [&](auto SET) {
    constexpr int S_ = decltype(SET)::value;
    hb[S_] = *lt_p;
    a1b[S_] = la_p[0];
    a2b[S_] = la_p[64];
    lt_p -= lt_p >= lt_0 + lt_back ? lt_back : 0;
    la_p -= la_p >= la_0 + la_back ? la_back : 0;
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  if (n > 0) {
    if ((n - 1) & 1) fetch(Set1{}); else fetch(Set0{});
  }
  if (n > 1) {
    if ((n - 2) & 1) fetch(Set1{}); else fetch(Set0{});
  }
  float dt_n = n > 0 ? sdt[klast] : 0.0f, t_n = n > 0 ? stt[klast] : 0.0f;
  auto euler_step = [&](auto SET, int s) {   // s: the step's index within the item
    constexpr int S_ = decltype(SET)::value;
    float hk = hb[S_], a1s = a1b[S_], a2s = a2b[S_];
    const float dt = dt_n, tdiff = t_n - tau_s;
    asm volatile("" : "+v"(hk), "+v"(a1s), "+v"(a2s));
    __builtin_amdgcn_sched_barrier(0);
    fetch(SET);
    dt_n = sdt[kbeg + (s > 0 ? s - 1 : 0)];
    t_n = stt[kbeg + (s > 0 ? s - 1 : 0)];
    __builtin_amdgcn_sched_barrier(0);
    const float th = cH ? tanh_f(hk) : 0.0f;
    *lm_p = lam;
    lm_p -= lm_back;
    float d3 = dt * lam;
    dpp_settle(d3);
    float gg = dpp_row_dot<H>(0.0f, d3, w3t);                       // W3^T delta3: unit layout
    float R[4];
    const float d2 = chain_delta<C::ACT, DROP>(gg, a2s, inv_keep, keepf);
    dpp_replicate(d2, R);
    gg = dpp_dot<W>(0.0f, R, w2t);                                 // W2^T delta2
    float d1 = chain_delta<C::ACT, DROP>(gg, a1s, inv_keep, keepf);
    ld_p[0] = d1;   // (for the pair dW kernel: no transposed product left in it)
    ld_p[64] = d2;   // (without records: 64 floats further into the scratch row)
    ld_p -= ld_back;
    s0acc += d1;
    s1acc = fmaf(d1, tdiff, s1acc);
    dpp_settle(d1);
    const float din = dpp_rows_sum(dpp_row_dot<NQW>(0.0f, d1, w1tp));   // W1h^T delta1, K-split
    lam = cH ? fmaf(din, 1.0f - th * th, lam) : 0.0f;
  };
  int s = n - 1;
  if (s >= 0 && (s & 1)) {
    euler_step(Set1{}, s);
    --s;
  }
  for (; s - 1 >= 0; s -= 2) {
    euler_step(Set0{}, s);
    euler_step(Set1{}, s - 1);
  }
  if (s >= 0) euler_step(Set0{}, s);
  if (cH && g == 0) a.lam_start[(size_t)r * H + c] = lam;
  if (a.cseg) {
    float* cs = a.cseg + (size_t)r * CHAIN_ACT_FLOATS + lane;
    cs[0] = s0acc;
    cs[64] = s1acc;
  }
}

}  // namespace njode
