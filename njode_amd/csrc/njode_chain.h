// njode_chain.h -- the lockstep plan of the masked (PhysioNet-shaped) models in the LATENCY regime:
// ONE WAVE PER PATH, a lane is a UNIT of the layer being evaluated.
//
// Why (round 6).  physionet_train.py:93 trains at B = 50; the reference's data set has 8 000
// patients (1 000 per GPU of an 8-GPU node).  A masked path is ONE serial chain of n_steps Euler
// steps (models.py:430-445) and ~n_obs jumps (models.py:457-489) -- at these batch sizes the step's
// time is the latency of that chain, not throughput.  The matrix-core kernels of njode_mfma_lock4.h
// run a chain as a 16-wide MFMA tile over four waves: 2.45 us per Euler step at ONE path per tile
// (15/16 of every matrix instruction is padding), three workgroup barriers and three LDS all-gathers
// per step (profiles/r06_step_critical_path.txt).  Here
//
//   * lane j of the wave owns unit j of every layer: row j of W1 (the state columns), W2, W3 of the
//     ODE network sits in that lane's REGISTERS (H + 2 W + 5 of them); the layer input -- 41 .. 50
//     floats -- is broadcast to all lanes from a wave-private LDS vector (ds_read_b128 of one
//     address) and consumed by ONE k-ordered fma chain per lane: H + 1 + 2 W fma per Euler step on
//     the wave's critical path, no matrix instruction, no workgroup barrier (LDS operations of one
//     wave execute in order), no cross-wave traffic;
//   * the x columns of W1 (the last prediction, constant between two jumps) and tau enter through a
//     per-segment accumulator c1 = b1 + W1x tanh(x) + w_tau tau, recomputed at a jump;
//   * the encoder and the readout (three evaluations per jump, ~65 jumps in 3 000 steps) read their
//     rows from LDS tables [quad][lane][4] shared by the block's waves (ds_read_b128, conflict free);
//   * everything per (path, step) is wave-uniform: the schedule, the jump bookkeeping, the dropout
//     decisions -- the 64-bit keep mask of a layer IS a lane mask (drawn ahead by k_chain_bits from
//     the streams of the matrix-core kernels: the same masks, bit for bit, which pass 2 of the
//     backward regenerates);
//   * a block is 1 .. 8 such waves (one path each; the only thing they share is the LDS tables), so
//     256 CUs hold 2 048 paths; above that the matrix-core tiles (throughput regime) take over.
//
// The kernels write exactly the buffers the kernels of njode_mfma_lock4.h write (ltraj, src_row,
// h_end, y_row, ybj_row, hT, loss_terms; lam_traj, g_y, g_ybj, g_hnew, g_hstart), so pass 2 of the
// backward (k_ode_dw_pairs_mfma, k_dec_dw_rows_mfma, k_enc_dw_rows_mfma) is unchanged; the stored
// hidden activations (lact / jact) have this file's own compact layout: [.][layer][64 lanes].
#pragma once
#include "njode_mfma_lock4.h"
#include "njode_dpp.h"

namespace njode {

constexpr int chain_q(int n) { return (n + 3) / 4; }

// (chain_step_bits / chain_row_bits / chain_masks: njode_mfma.h, beside the streams they re-assemble)

template <class C> __global__ void __launch_bounds__(256) k_chain_bits(KArgs a) {
  uint64_t* sb = (uint64_t*)a.dbits;
  uint64_t* rb = (uint64_t*)a.dbits_row;
  const long long n_ode = (long long)a.K * a.B, n_row = (long long)a.n_obs * 3 + a.B;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_ode + n_row;
       i += (long long)gridDim.x * 256) {
    uint64_t m1, m2;
    if (i < n_ode) {
      const int b = (int)(i / a.K), k = (int)(i % a.K);
      chain_masks<MF<C>::Q1>(a.dc, a.gid0 + b, (uint32_t)k, NET_ODE, m1, m2);
      sb[i * 2] = m1;
      sb[i * 2 + 1] = m2;
    } else {
      const long long j = i - n_ode;
      if (j < (long long)a.n_obs * 3) {
        const int row = (int)(j / 3), e = (int)(j % 3);
        const uint32_t net = e == 0 ? NET_DEC_BJ : (e == 1 ? NET_ENC : NET_DEC);
        chain_masks<16>(a.dc, a.gid0 + a.obs_idx[row], (uint32_t)a.k_jump[a.t_of_row[row]], net, m1, m2);
      } else {
        const int b = (int)(j - (long long)a.n_obs * 3);
        chain_masks<16>(a.dc, a.gid0 + b, TKEY_START, NET_ENC, m1, m2);
      }
      rb[j * 2] = m1;
      rb[j * 2 + 1] = m2;
    }
  }
}

// ---- LDS tables [quad q][lane][4]: entry k = 4 q + e of lane l is W[unit(l) * ld_unit + k * ld_k] -------
// (unit(l) = dpp_unit(l), njode_dpp.h; rows of units >= n_units and entries k >= NK are zero)
template <int NK>
NJ_DEV void chain_fill(lfp tab, const float* __restrict__ Wp, int n_units, int ld_unit, int ld_k, int tid,
                       int nthreads) {
  constexpr int NQ = chain_q(NK);
  for (int i = tid; i < NQ * 256; i += nthreads) {
    const int e = i & 3, u = dpp_unit((i >> 2) & 63), k = 4 * (i >> 8) + e;
    tab[i] = (u < n_units && k < NK) ? Wp[(size_t)u * ld_unit + (size_t)k * ld_k] : 0.0f;
  }
}
// acc + sum_k tab[lane][k] unit(k): the lane's row comes from the LDS table in blocks of four quads,
// one block ahead of the fma chain that consumes it (the reads depend on nothing the chain
// computes); the padding entries of the last quad are zero, so every quad is a full one
template <int NQ, int Q> NJ_DEV void chain_lds_blocks(float& acc0, float& acc1, lfp tab_lane, const float (&R)[4], const f4 (&cur)[4]) {
  constexpr int N = NQ - Q >= 4 ? 4 : NQ - Q;
  f4 nxt[4] = {cur[0], cur[0], cur[0], cur[0]};
  if constexpr (Q + 4 < NQ) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (Q + 4 + j < NQ) nxt[j] = *(lf4p)(tab_lane + (Q + 4 + j) * 256);
  }
  __builtin_amdgcn_sched_barrier(0);
  float w[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    w[4 * j] = cur[j].x;
    w[4 * j + 1] = cur[j].y;
    w[4 * j + 2] = cur[j].z;
    w[4 * j + 3] = cur[j].w;
  }
  dpp_block<Q, N>(acc0, R, w);
  if constexpr (Q + 4 < NQ) chain_lds_blocks<NQ, Q + 4>(acc1, acc0, tab_lane, R, nxt);   // (two accumulators: njode_dpp.h)
}
template <int NK> NJ_DEV float chain_dot_lds(lfp tab_lane, const float (&R)[4], float init) {
  constexpr int NQ = chain_q(NK);
  static_assert(NQ <= 16, "one input vector is at most 64 units");
  f4 cur[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) cur[j] = *(lf4p)(tab_lane + (j < NQ ? j : 0) * 256);
  float acc0 = init, acc1 = 0.0f;
  chain_lds_blocks<NQ, 0>(acc0, acc1, tab_lane, R, cur);
  return NQ > 4 ? acc0 + acc1 : acc0;
}
template <int NK>
NJ_DEV void chain_load_row(float (&w)[NK], const float* __restrict__ Wp, bool on, int ld_k) {
#pragma unroll
  for (int k = 0; k < NK; ++k) w[k] = on ? Wp[(size_t)k * ld_k] : 0.0f;
}
// sum of the vector whose own unit is v (units >= N hold zero), the same value in every lane: the
// dot product with ones (dpp_dot's fixed summation order)
template <int N> NJ_DEV float chain_sum(float v) {
  float R[4], ones[N];
  dpp_replicate(v, R);
#pragma unroll
  for (int k = 0; k < N; ++k) ones[k] = 1.0f;
  return dpp_dot<N>(0.0f, R, ones);
}

// A wave-uniform value that sits in a vector register (loaded through a plain pointer) into a scalar
// one.  Inline assembly on purpose: the compiler KNOWS the value is uniform, folds the builtin away
// and then, having it in a vector register, turns every loop and branch that depends on it into
// exec-masked vector code (and the scalar loads indexed by it into vector loads).
NJ_DEV int chain_sgpr(int v) {
  int r;
  // (the s_nop in FRONT: a cross-lane read of a register the preceding VALU instruction wrote needs
  // wait states that the hazard recognizer does not insert for inline assembly -- without them the
  // scalar register received garbage on gfx950)
  asm volatile("s_nop 4\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(r) : "v"(v));
  return r;
}

// hidden activation of the lane's unit; `keep` is the layer's 64-bit LANE mask (wave-uniform, in
// scalar registers: the select is one v_cndmask on it).  A dropped unit is -0.0f (the mark the
// sweep reads back: a kept unit is never -0, the fma with +0 sees to that)
template <int ACT, bool DROP> NJ_DEV float chain_hidden(float z, uint64_t keep, float inv_keep) {
  float v = act_f<ACT>(z);
  if constexpr (DROP) {
    const float kv = fmaf(v, inv_keep, 0.0f), nz = -0.0f;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(v) : "v"(nz), "v"(kv), "s"(keep));
  }
  return v;
}
template <int ACT, bool DROP> NJ_DEV float chain_delta(float g, float a, float inv_keep, float keepf) {
  if constexpr (DROP) return __float_as_uint(a) != 0x80000000u ? g * inv_keep * dact_f<ACT>(a * keepf) : 0.0f;
  else return g * dact_f<ACT>(a);
}

template <class C> struct ChainLds {
  static constexpr int D = C::D, H = C::H, W = C::W;
  // forward tables: encoder W1 (value columns | mask columns) W2 W3, readout W1 W2 W3, the x
  // columns of the ODE network's W1
  static constexpr int FE1 = 0, FE1M = FE1 + chain_q(D) * 256, FE2 = FE1M + (C::MASKED ? chain_q(D) * 256 : 0),
                       FE3 = FE2 + chain_q(W) * 256, FD1 = FE3 + chain_q(W) * 256, FD2 = FD1 + chain_q(H) * 256,
                       FD3 = FD2 + chain_q(W) * 256, FX1 = FD3 + chain_q(W) * 256,
                       FWD_FLOATS = FX1 + chain_q(D) * 256;
  // sweep tables (transposed products): readout W3^T W2^T W1^T, encoder W3^T W2^T W1x^T, ODE W1x^T
  static constexpr int BD3 = 0, BD2 = BD3 + chain_q(C::DO) * 256, BD1 = BD2 + chain_q(W) * 256,
                       BE3 = BD1 + chain_q(W) * 256, BE2 = BE3 + chain_q(H) * 256, BE1 = BE2 + chain_q(W) * 256,
                       BX1 = BE1 + chain_q(W) * 256, BWD_FLOATS = BX1 + chain_q(W) * 256;
  static_assert(FWD_FLOATS * 4 <= 160 * 1024 && BWD_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

#ifdef NJ_CHAIN_STAMP
#define CH_STAMP_DECL unsigned long long ch_ts[12]; int ch_nts = 0; (void)ch_ts; (void)ch_nts
#define CH_STAMP() do { if (ch_on && ch_nts < 12) ch_ts[ch_nts++] = __builtin_readcyclecounter(); } while (0)
#define CH_STAMP_PRINT(name) do { if (ch_on && lane == 0) { \
    unsigned long long d_[10]; for (int i_ = 0; i_ < 10; ++i_) d_[i_] = i_ + 1 < ch_nts ? ch_ts[i_ + 1] - ch_ts[i_] : 0; \
    printf("%s: %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu | total %llu\n", name, d_[0], d_[1], d_[2], d_[3], \
           d_[4], d_[5], d_[6], d_[7], d_[8], d_[9], ch_ts[ch_nts - 1] - ch_ts[0]); } ch_nts = 0; } while (0)
#else
#define CH_STAMP_DECL
#define CH_STAMP()
#define CH_STAMP_PRINT(name)
#endif

// =====================================================================================
// forward
// =====================================================================================
template <class C, bool DROP>
__global__ void __launch_bounds__(64 * CHAIN_MAX_WAVES) k_paths_fwd_chain(KArgs a) {
  using NO = typename C::Ode;
  using NE = typename C::Enc;
  using ND = typename C::Dec;
  using L = ChainLds<C>;
  constexpr int D = C::D, H = C::H, DO = C::DO, W = C::W, EIN = C::ENC_IN, OIN = C::ODE_IN;
  __shared__ __attribute__((aligned(16))) float lds_raw[L::FWD_FLOATS];
  lfp T = (lfp)lds_raw;
  const int lane = threadIdx.x & 63, u = dpp_unit(lane);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = (int)(blockDim.x >> 6);
  const float* Po = a.P + C::OFF_ODE;
  const float* Pe = a.P + C::OFF_ENC;
  const float* Pd = a.P + C::OFF_DEC;
  {
    const int tid = threadIdx.x, nt = blockDim.x;
    chain_fill<D>(T + L::FE1, Pe + NE::woff(0), W, EIN, 1, tid, nt);
    if constexpr (C::MASKED) chain_fill<D>(T + L::FE1M, Pe + NE::woff(0) + D, W, EIN, 1, tid, nt);
    chain_fill<W>(T + L::FE2, Pe + NE::woff(1), W, W, 1, tid, nt);
    chain_fill<W>(T + L::FE3, Pe + NE::woff(2), H, W, 1, tid, nt);
    chain_fill<H>(T + L::FD1, Pd + ND::woff(0), W, H, 1, tid, nt);
    chain_fill<W>(T + L::FD2, Pd + ND::woff(1), W, W, 1, tid, nt);
    chain_fill<W>(T + L::FD3, Pd + ND::woff(2), DO, W, 1, tid, nt);
    chain_fill<D>(T + L::FX1, Po + NO::woff(0), W, OIN, 1, tid, nt);   // (x columns: 0 .. D-1)
  }
  __syncthreads();
  const int b = blockIdx.x * wpb + wv;   // (wave-uniform)
  if (b >= a.B) return;

  lfp tE1 = T + L::FE1 + lane * 4, tE1M = T + L::FE1M + lane * 4, tE2 = T + L::FE2 + lane * 4,
      tE3 = T + L::FE3 + lane * 4, tD1 = T + L::FD1 + lane * 4, tD2 = T + L::FD2 + lane * 4,
      tD3 = T + L::FD3 + lane * 4, tX1 = T + L::FX1 + lane * 4;

  // the ODE network's rows in registers: state columns of W1, tau / tdiff (/ t) columns, W2, W3
  float w1h[H], w2[W], w3[W];
  const bool uW = u < W, uH = u < H, uD = u < D, uO = u < DO;
  const int jW = uW ? u : 0, jH = uH ? u : 0, jD = uD ? u : 0, jO = uO ? u : 0;
  chain_load_row<H>(w1h, Po + NO::woff(0) + (size_t)jW * OIN + D, uW, 1);
  chain_load_row<W>(w2, Po + NO::woff(1) + (size_t)jW * W, uW, 1);
  chain_load_row<W>(w3, Po + NO::woff(2) + (size_t)jH * W, uH, 1);
  const float w1tau = uW ? Po[NO::woff(0) + (size_t)jW * OIN + D + H] : 0.0f;
  const float w1td = uW ? Po[NO::woff(0) + (size_t)jW * OIN + D + H + 1] : 0.0f;
  const float w1ct = (C::CURT && uW) ? Po[NO::woff(0) + (size_t)jW * OIN + D + H + 2] : 0.0f;
  const float ob1 = uW ? Po[NO::boff(0) + jW] : 0.0f, ob2 = uW ? Po[NO::boff(1) + jW] : 0.0f,
              ob3 = uH ? Po[NO::boff(2) + jH] : 0.0f;
  const float eb1 = uW ? Pe[NE::boff(0) + jW] : 0.0f, eb2 = uW ? Pe[NE::boff(1) + jW] : 0.0f,
              eb3 = uH ? Pe[NE::boff(2) + jH] : 0.0f;
  const float db1 = uW ? Pd[ND::boff(0) + jW] : 0.0f, db2 = uW ? Pd[ND::boff(1) + jW] : 0.0f,
              db3 = uO ? Pd[ND::boff(2) + jO] : 0.0f;

  const bool LOSS = a.want_loss != 0, SAVE = a.save_traj != 0;
  const int __attribute__((address_space(4)))* kjump =
      (const int __attribute__((address_space(4)))*)(unsigned long long)a.k_jump;
  const int __attribute__((address_space(4)))* t_of_row =
      (const int __attribute__((address_space(4)))*)(unsigned long long)a.t_of_row;
  const int __attribute__((address_space(4)))* row_by_path =
      (const int __attribute__((address_space(4)))*)(unsigned long long)a.row_by_path;
  const int __attribute__((address_space(4)))* path_sorted =
      (const int __attribute__((address_space(4)))*)(unsigned long long)a.path_sorted;
  // (wave-uniform data through the scalar cache: the Euler step then has NO vector load and no LDS
  // operation -- nothing for its stores to be waited behind, no s_waitcnt but the scalar one)
  typedef const unsigned long long __attribute__((address_space(4)))* cu64p;
  const cu64p sbits = (cu64p)(unsigned long long)a.dbits;
  const cu64p rbits = (cu64p)(unsigned long long)a.dbits_row;
  const cfp sdt = as_cfp(a.step_dt), stt = as_cfp(a.step_t), tf32 = as_cfp(a.time_f32);
  const float inv_keep = a.dc.inv_keep;

  // layers 2 and 3 of a network whose rows come from the LDS tables; a1o: the first hidden
  // activation of the own unit (in), a2o: the second (out)
  auto net_tail = [&](lfp t2, lfp t3, float a1o, float bb2, float bb3, uint64_t m2, float& a2o) {
    float R[4];
    dpp_replicate(a1o, R);
    const float z = chain_dot_lds<W>(t2, R, bb2);
    a2o = chain_hidden<C::ACT, DROP>(z, m2, inv_keep);
    dpp_replicate(a2o, R);
    return chain_dot_lds<W>(t3, R, bb3);
  };
  // readout of the state whose tanh (own unit) is thq
  auto readout = [&](float hq, float thq, uint64_t m1, uint64_t m2, float& a1o, float& a2o) {
    float R[4];
    dpp_replicate(thq, R);
    const float z = chain_dot_lds<H>(tD1, R, db1);
    a1o = chain_hidden<C::ACT, DROP>(z, m1, inv_keep);
    float y = net_tail(tD2, tD3, a1o, db2, db3, m2, a2o);
    if constexpr (C::DEC_CASE == 1) y += hq;
    return uO ? y : 0.0f;
  };
  // encoder of [tanh(xin), mask] (own units txin, m)
  auto encode = [&](float xin, float txin, float m, uint64_t m1, uint64_t m2, float& a1o, float& a2o) {
    float R[4];
    dpp_replicate(txin, R);
    float z = chain_dot_lds<D>(tE1, R, eb1);
    if constexpr (C::MASKED) {
      dpp_replicate(m, R);
      z = chain_dot_lds<D>(tE1M, R, z);
    }
    a1o = chain_hidden<C::ACT, DROP>(z, m1, inv_keep);
    float hq = net_tail(tE2, tE3, a1o, eb2, eb3, m2, a2o);
    if constexpr (C::ENC_CASE == 1) hq += xin;
    return uH ? hq : 0.0f;
  };
  // the segment's constant part of the ODE network's first layer: b1 + W1x tanh(x) + w_tau tau
  auto segment_c1 = [&](float txq, float tauq) {
    float R[4];
    dpp_replicate(txq, R);
    return fmaf(w1tau, tauq, chain_dot_lds<D>(tX1, R, ob1));
  };

  // ---- initial state: h = encoder(start_X, mask = 0) (models.py:404-413) ---------------------------
  const float xs = uD ? a.start_X[(size_t)b * D + jD] : 0.0f;
  float tx = tanh_f(xs);   // tanh(last_X), own unit (0 for the padding units: xs = 0)
  float h, th;
  {
    uint64_t m1 = 0, m2 = 0;
    if constexpr (DROP) {
      m1 = rbits[chain_row_bits(a.n_obs, 0) + (size_t)b * 2];
      m2 = rbits[chain_row_bits(a.n_obs, 0) + (size_t)b * 2 + 1];
    }
    float a1l, a2l;
    h = encode(xs, tx, 0.0f, m1, m2, a1l, a2l);
    th = tanh_f(h);
  }
  float tau = 0.0f;
  float c1 = segment_c1(tx, tau);

  // the path's next observation (wave-uniform): index into the path-sorted rows, row, Euler step
  int cur = chain_sgpr(a.first_j[b]);   // (the loops on it are scalar loops)
  int r_next = -1, i_next = 0, k_next = 0x7fffffff;
  auto load_next = [&]() {
    const bool on = a.n_obs > 0 && cur >= 0 && cur < a.n_obs && path_sorted[cur < a.n_obs && cur >= 0 ? cur : 0] == b;
    if (on) {
      r_next = row_by_path[cur];
      i_next = t_of_row[r_next];
      k_next = kjump[i_next];
    } else {
      r_next = -1;
      k_next = 0x7fffffff;
    }
  };
  load_next();
  int src = -1;
  float loss_acc = 0.0f;

  // the per-step stores are unconditional (a lane / a call that saves nothing stores to `trash`)
  float* const trash = a.trash + threadIdx.x;
  float* lt_p = (SAVE && uH) ? a.ltraj + (size_t)b * H + jH : trash;
  const size_t lt_step = (SAVE && uH) ? (size_t)a.B * H : 0;
  float* la_p = SAVE ? a.lact + (size_t)b * CHAIN_ACT_FLOATS + lane : trash;
  const size_t la_step = SAVE ? (size_t)a.B * CHAIN_ACT_FLOATS : 0;
  const int la_2 = SAVE ? 64 : 0;
  int* sr_p = (SAVE && lane == 0) ? a.src_row + b : (int*)trash;
  const size_t sr_step = (SAVE && lane == 0) ? (size_t)a.B : 0;

  // the schedule's values and the keep masks of a step are loaded one step ahead
  float dt_n = a.K > 0 ? sdt[0] : 0.0f, t_n = a.K > 0 ? stt[0] : 0.0f;
  uint64_t m1_n = 0, m2_n = 0;
  if constexpr (DROP) {
    if (a.K > 0) {
      m1_n = sbits[chain_step_bits(0, a.K, b)];
      m2_n = sbits[chain_step_bits(0, a.K, b) + 1];
    }
  }
  float y_cur = 0.0f;   // the current prediction, own unit (prediction calls: path_y)
  auto euler_step = [&](int k) {   // ---- Euler step k (models.py:369-377, 430-445)
    *lt_p = h;   // state before step k
    lt_p += lt_step;
    *sr_p = src;
    sr_p += sr_step;
    // ---- Euler step k (models.py:369-377, 430-445)
    const float dt = dt_n, t = t_n;
    const uint64_t m1 = m1_n, m2 = m2_n;
    {   // (unconditional, clamped index: a load under `if (k + 1 < K)` makes the loop-carried value
        // a vector register, and the copy waits for the scalar load where it is issued)
      const int kn = k + 1 < a.K ? k + 1 : k;
      dt_n = sdt[kn];
      t_n = stt[kn];
      if constexpr (DROP) {
        m1_n = sbits[chain_step_bits(kn, a.K, b)];
        m2_n = sbits[chain_step_bits(kn, a.K, b) + 1];
      }
    }
#ifdef NJ_CHAIN_STAMP
    const bool ch_on = b == 0 && k == a.K / 2;
#endif
    CH_STAMP_DECL;
    CH_STAMP();
    float R[4];
    float z = fmaf(w1td, t - tau, c1);
    if constexpr (C::CURT) z = fmaf(w1ct, tau + (t - tau), z);
    dpp_replicate(th, R);
    z = dpp_dot<H>(z, R, w1h);
    CH_STAMP();
    const float a1l = chain_hidden<C::ACT, DROP>(z, m1, inv_keep);
    dpp_replicate(a1l, R);
    CH_STAMP();
    z = dpp_dot<W>(ob2, R, w2);
    CH_STAMP();
    const float a2l = chain_hidden<C::ACT, DROP>(z, m2, inv_keep);
    dpp_replicate(a2l, R);
    CH_STAMP();
    const float f = dpp_dot<W>(ob3, R, w3);
    CH_STAMP();
    la_p[0] = a1l;
    la_p[la_2] = a2l;
    la_p += la_step;
    h = uH ? fmaf(dt, f, h) : 0.0f;
    th = tanh_f(h);
    CH_STAMP();
    CH_STAMP_PRINT("chain fwd");
  };
  auto jump = [&]() {   // ---- jump (models.py:457-489): this path observes before step k_next
    const int r_ = r_next;
    const float xr = a.X[(size_t)r_ * D + jD];
    const float mr = C::MASKED ? a.M[(size_t)r_ * D + jD] : 1.0f;
    const float tnew = tf32[i_next];
    uint64_t jm[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    if constexpr (DROP) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        jm[e][0] = rbits[chain_row_bits(r_, e)];
        jm[e][1] = rbits[chain_row_bits(r_, e) + 1];
      }
    }
    float* ja_p = a.jact + (size_t)r_ * CHAIN_JACT_FLOATS + lane;
    // the path's next row, loaded beside the jump
    ++cur;
    load_next();
    if (SAVE && uH) a.h_end[(size_t)r_ * H + u] = h;
    float a1l, a2l;
    const float ybj = readout(h, th, jm[0][0], jm[0][1], a1l, a2l);   // y_bj = readout(h)
    if (SAVE) {
      ja_p[0] = a1l;
      ja_p[64] = a2l;
    }
    const float x = uD ? xr : 0.0f, m = uD ? mr : 0.0f;
    const float xin = C::MASKED ? x * m + (1.0f - m) * ybj : x;
    const float hn = encode(xin, uD ? tanh_f(xin) : 0.0f, m, jm[1][0], jm[1][1], a1l, a2l);
    if (SAVE) {
      ja_p[128] = a1l;
      ja_p[192] = a2l;
    }
    const float thn = tanh_f(hn);
    const float yn = readout(hn, thn, jm[2][0], jm[2][1], a1l, a2l);
    if (SAVE) {
      ja_p[256] = a1l;
      ja_p[320] = a2l;
      if (uO) {
        a.y_row[(size_t)r_ * DO + u] = yn;
        a.ybj_row[(size_t)r_ * DO + u] = ybj;
      }
    }
    if (LOSS) {   // compute_loss (models.py:76-110) of this row
      const float e = x - yn;
      const float f = a.loss_easy ? (ybj - x) : (ybj - yn);
      const float sa = chain_sum<D>(m * e * e);
      const float sb = chain_sum<D>(m * f * f);
      const float scale = a.inv_batch * __builtin_amdgcn_rcpf((float)a.n_obs_ot[b]);
      const float na = sqrtf(sa + 1e-10f), nb = sqrtf(sb + 1e-10f);
      const float ca = a.loss_easy ? a.weight : 2.0f * a.weight;
      const float cb = a.loss_easy ? (1.0f - a.weight) : 2.0f * (1.0f - a.weight);
      const float s = ca * na + cb * nb;
      loss_acc += s * s * scale;
    }
    // commit (models.py:463-489)
    h = hn;
    th = thn;
    tx = uD ? tanh_f(C::MASKED ? yn : x) : 0.0f;
    tau = tnew;
    src = r_;
    c1 = segment_c1(tx, tau);
    y_cur = yn;
  };
  if (a.want_path) {
    // Prediction calls (return_path; evaluate / get_pred): the batch's LOCKSTEP row structure -- a row
    // of (h, readout(h)) at t = 0, behind every Euler step and at every observation TIME of the batch,
    // whether this path observes then or not (models.py:423-426, 442-445, 491-494) -- so the times are
    // walked one by one and the readout runs after every step.  (Dropout-free calls only: njode_cfg.hip.)
    const int DOp = DO;
    int row = 0, i_all = 0;
    auto write_row = [&]() {
      if (uH) a.path_h[((size_t)row * a.B + b) * H + u] = h;
      if (uO) a.path_y[((size_t)row * a.B + b) * DOp + u] = y_cur;
      ++row;
    };
    {
      float a1l, a2l;
      y_cur = readout(h, th, 0, 0, a1l, a2l);
    }
    write_row();
    for (int k = 0;; ++k) {
      while (i_all < a.n_times && kjump[i_all] == k) {
        if (k_next == k && i_next == i_all) jump();
        write_row();
        ++i_all;
      }
      if (k >= a.K) break;
      euler_step(k);
      float a1l, a2l;
      y_cur = readout(h, th, 0, 0, a1l, a2l);
      write_row();
    }
  } else {
    // The path's chain, segment by segment: the Euler steps up to the next observation are a plain
    // counted loop (nothing in it but the step: its scalar prefetches and stores keep one shape), then
    // the jump.  (Two observation times closer than 1e-10 dt share a step index: a segment of no steps.)
    int k = 0;
    for (;;) {
      const int k_stop = k_next < a.K ? k_next : a.K;
      for (; k < k_stop; ++k) euler_step(k);
      if (k_next > a.K) break;
      jump();
    }
  }
  *lt_p = h;   // the final state: ltraj[K]
  if (uH) a.hT[(size_t)b * H + u] = h;
  if (LOSS && lane == 0) a.loss_terms[b] = loss_acc;
}

// =====================================================================================
// adjoint sweep (pass 1 of the backward)
// =====================================================================================
template <class C, bool DROP>
__global__ void __launch_bounds__(64 * CHAIN_MAX_WAVES) k_paths_bwd_adj_chain(KArgs a) {
  using NO = typename C::Ode;
  using NE = typename C::Enc;
  using ND = typename C::Dec;
  using L = ChainLds<C>;
  constexpr int D = C::D, H = C::H, DO = C::DO, W = C::W, EIN = C::ENC_IN, OIN = C::ODE_IN;
  __shared__ __attribute__((aligned(16))) float lds_raw[L::BWD_FLOATS];
  lfp T = (lfp)lds_raw;
  const int lane = threadIdx.x & 63, u = dpp_unit(lane);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = (int)(blockDim.x >> 6);
  const float* Po = a.P + C::OFF_ODE;
  const float* Pe = a.P + C::OFF_ENC;
  const float* Pd = a.P + C::OFF_DEC;
  {
    const int tid = threadIdx.x, nt = blockDim.x;
    // lane = input unit of the layer, entry k = output unit: W[k][unit]
    chain_fill<DO>(T + L::BD3, Pd + ND::woff(2), W, 1, W, tid, nt);
    chain_fill<W>(T + L::BD2, Pd + ND::woff(1), W, 1, W, tid, nt);
    chain_fill<W>(T + L::BD1, Pd + ND::woff(0), H, 1, H, tid, nt);
    chain_fill<H>(T + L::BE3, Pe + NE::woff(2), W, 1, W, tid, nt);
    chain_fill<W>(T + L::BE2, Pe + NE::woff(1), W, 1, W, tid, nt);
    chain_fill<W>(T + L::BE1, Pe + NE::woff(0), D, 1, EIN, tid, nt);   // (the tanh(x) inputs only)
    chain_fill<W>(T + L::BX1, Po + NO::woff(0), D, 1, OIN, tid, nt);   // (x columns of the ODE network)
  }
  __syncthreads();
  const int b = blockIdx.x * wpb + wv;
  if (b >= a.B) return;

  lfp tD3 = T + L::BD3 + lane * 4, tD2 = T + L::BD2 + lane * 4, tD1 = T + L::BD1 + lane * 4,
      tE3 = T + L::BE3 + lane * 4, tE2 = T + L::BE2 + lane * 4, tE1 = T + L::BE1 + lane * 4,
      tX1 = T + L::BX1 + lane * 4;

  const bool uW = u < W, uH = u < H, uD = u < D, uO = u < DO;
  const int jW = uW ? u : 0, jH = uH ? u : 0, jD = uD ? u : 0;
  // transposed rows of the ODE network in registers: W3[.][j], W2[.][j], W1[.][state column u]
  float w3t[H], w2t[W], w1t[W];
  chain_load_row<H>(w3t, Po + NO::woff(2) + jW, uW, W);
  chain_load_row<W>(w2t, Po + NO::woff(1) + jW, uW, W);
  chain_load_row<W>(w1t, Po + NO::woff(0) + D + jH, uH, OIN);

  const int __attribute__((address_space(4)))* kjump =
      (const int __attribute__((address_space(4)))*)(unsigned long long)a.k_jump;
  const int __attribute__((address_space(4)))* t_of_row =
      (const int __attribute__((address_space(4)))*)(unsigned long long)a.t_of_row;
  const int __attribute__((address_space(4)))* item_prev =
      (const int __attribute__((address_space(4)))*)(unsigned long long)a.item_prev;
  const float inv_keep = a.dc.inv_keep, keepf = a.keep;

  float lam = (a.g_hT && uH) ? a.g_hT[(size_t)b * H + jH] : 0.0f;   // adjoint of h, own unit
  float d1acc = 0.0f;   // sum of delta1 over the steps of the segment, own unit
  // ... and of delta1 (t - tau), tau = the time of the observation in front of the segment: the x / tau /
  // time columns of dW1 are one outer product per SEGMENT (njode_chain_dw.h)
  float s1acc = 0.0f, tau_s = 0.0f;

  // the row this path reverses next and the Euler step in front of which its jump sits
  int src = chain_sgpr(a.last_row[b]), src_k = -1;   // (scalar loops)
  if (src >= 0) src_k = kjump[t_of_row[src]];

  // adjoint of y = readout(hq) w.r.t. hq: dy (own unit) -> dh (own unit); a1s / a2s: the
  // evaluation's hidden activations of the own unit as the forward stored them
  auto dec_adj = [&](float hq, float dy, float a1s, float a2s) {
    const float thq = tanh_f(hq);
    float R[4];
    dpp_replicate(dy, R);
    float g = chain_dot_lds<DO>(tD3, R, 0.0f);
    dpp_replicate(chain_delta<C::ACT, DROP>(g, a2s, inv_keep, keepf), R);
    g = chain_dot_lds<W>(tD2, R, 0.0f);
    dpp_replicate(chain_delta<C::ACT, DROP>(g, a1s, inv_keep, keepf), R);
    const float din = chain_dot_lds<W>(tD1, R, 0.0f);
    float v = din * (1.0f - thq * thq);
    if constexpr (C::DEC_CASE == 1) v += dy;
    return uH ? v : 0.0f;
  };

  const float* lt_p = a.ltraj + ((size_t)(a.K > 0 ? a.K - 1 : 0) * a.B + b) * H + jH;
  float* lm_p = uH ? a.lam_traj + ((size_t)(a.K > 0 ? a.K - 1 : 0) * a.B + b) * H + jH : a.trash + threadIdx.x;
  // delta1 | delta2 of every step, for the pair dW kernel (a.cdelta null: it recomputes them)
  float* ld_p = a.cdelta ? a.cdelta + ((size_t)(a.K > 0 ? a.K - 1 : 0) * a.B + b) * CHAIN_ACT_FLOATS + lane
                         : a.trash + threadIdx.x;
  const size_t ld_back = a.cdelta ? (size_t)a.B * CHAIN_ACT_FLOATS : 0;
  const size_t lt_back = (size_t)a.B * H, lm_back = uH ? lt_back : 0;
  const cfp sdt = as_cfp(a.step_dt), stt = as_cfp(a.step_t), tf32 = as_cfp(a.time_f32);
  float* const cs_base = a.cseg ? a.cseg + lane : a.trash + threadIdx.x;
  const size_t cs_rec = a.cseg ? CHAIN_ACT_FLOATS : 0;
  const float* la_p = a.lact + ((size_t)(a.K > 0 ? a.K - 1 : 0) * a.B + b) * CHAIN_ACT_FLOATS + lane;
  const size_t la_back = (size_t)a.B * CHAIN_ACT_FLOATS;
  const float* const lt_0 = a.ltraj + (size_t)b * H + jH;
  const float* const la_0 = a.lact + (size_t)b * CHAIN_ACT_FLOATS + lane;
  // Two steps ahead (a step is shorter than a trip to HBM): step k consumes set k & 1 and refills it
  // with the data of step k - 2; the set index is a compile-time constant of the two instances of the
  // step body, so there are no register copies that would wait for the loads where they are issued.
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
  if (a.K > 0) {
    if ((a.K - 1) & 1) fetch(Set1{}); else fetch(Set0{});
  }
  if (a.K > 1) {
    if ((a.K - 2) & 1) fetch(Set1{}); else fetch(Set0{});
  }
  float h_cur = 0.0f;
  float dt_n = a.K > 0 ? sdt[a.K - 1] : 0.0f;   // (scalar cache, one step ahead)
  float t_n = a.K > 0 ? stt[a.K - 1] : 0.0f;

  // ---- reverse Euler step k (the forward's hidden activations: no recompute)
  auto euler_step = [&](auto SET, int k) {
    constexpr int S_ = decltype(SET)::value;
    float hk = hb[S_], a1s = a1b[S_], a2s = a2b[S_];
    const float dt = dt_n, tdiff = t_n - tau_s;
    // the set is consumed HERE, in front of the loads that refill it: whatever wait the compiler
    // places for it then covers loads that are at least one step old, never the ones issued below
    asm volatile("" : "+v"(hk), "+v"(a1s), "+v"(a2s));
    __builtin_amdgcn_sched_barrier(0);
    fetch(SET);   // (unconditional: steps 1 and 0 re-read step 0's record)
    dt_n = sdt[k > 0 ? k - 1 : 0];
    t_n = stt[k > 0 ? k - 1 : 0];
    __builtin_amdgcn_sched_barrier(0);
#ifdef NJ_CHAIN_STAMP
    const bool ch_on = b == 0 && k == a.K / 2;
#endif
    CH_STAMP_DECL;
    CH_STAMP();
    h_cur = hk;
    const float th = uH ? tanh_f(hk) : 0.0f;
    *lm_p = lam;
    lm_p -= lm_back;
    float R[4];
    dpp_replicate(dt * lam, R);
    CH_STAMP();
    float g = dpp_dot<H>(0.0f, R, w3t);
    CH_STAMP();
    const float d2 = chain_delta<C::ACT, DROP>(g, a2s, inv_keep, keepf);
    dpp_replicate(d2, R);
    CH_STAMP();
    g = dpp_dot<W>(0.0f, R, w2t);
    CH_STAMP();
    const float d1 = chain_delta<C::ACT, DROP>(g, a1s, inv_keep, keepf);
    ld_p[0] = d1;
    ld_p[64] = d2;   // (without records: 64 floats further into the scratch row)
    ld_p -= ld_back;
    d1acc += d1;
    s1acc = fmaf(d1, tdiff, s1acc);
    dpp_replicate(d1, R);
    CH_STAMP();
    const float din = dpp_dot<W>(0.0f, R, w1t);
    lam = uH ? fmaf(din, 1.0f - th * th, lam) : 0.0f;
    CH_STAMP();
    CH_STAMP_PRINT("chain bwd");
  };

  // The chain in reverse, segment by segment: the steps down to the jump in front of step src_k are a
  // counted loop of PAIRS (even step on set 0, odd step on set 1 -- straight-line code, so the loads a
  // step issues into the set it has just consumed land in the loop-carried registers themselves and
  // every wait is a counted one), then the jump is reversed.
  float h_after = (uH && a.K >= 0) ? a.ltraj[((size_t)a.K * a.B + b) * H + jH] : 0.0f;   // state after the jump reversed next
  int k = a.K;   // steps k .. K-1 are reversed
  for (;;) {
    const int k_stop = src >= 0 ? src_k : 0;
    tau_s = src >= 0 ? tf32[t_of_row[src]] : 0.0f;
    int kk = k - 1;
    if (kk >= k_stop && (kk & 1)) {
      euler_step(Set1{}, kk);
      --kk;
    }
    for (; kk - 1 >= k_stop; kk -= 2) {
      euler_step(Set0{}, kk);
      euler_step(Set1{}, kk - 1);
    }
    if (kk >= k_stop) euler_step(Set0{}, kk);
    if (k > k_stop) h_after = h_cur;   // (the state before step k_stop)
    k = k_stop;
    {   // the segment's sums: behind row src, or behind the path's start
      float* cs = cs_base + (size_t)(src >= 0 ? src : a.n_obs + b) * cs_rec;
      cs[0] = d1acc;
      cs[64] = s1acc;
      s1acc = 0.0f;
    }
    if (src < 0) break;
    {   // ---- reverse the jump applied right before step k
      const int r_ = src;
      const float hn = h_after;
      const float hp = uH ? a.h_end[(size_t)r_ * H + jH] : 0.0f;
      const float x = uD ? a.X[(size_t)r_ * D + jD] : 0.0f;
      const float m = uD ? (C::MASKED ? a.M[(size_t)r_ * D + jD] : 1.0f) : 0.0f;
      const float y = uD ? a.y_row[(size_t)r_ * DO + jD] : 0.0f;
      const float ybj = uD ? a.ybj_row[(size_t)r_ * DO + jD] : 0.0f;
      const float* ja_p = a.jact + (size_t)r_ * CHAIN_JACT_FLOATS + lane;
      float ja[3][2];
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        ja[e][0] = ja_p[e * 128];
        ja[e][1] = ja_p[e * 128 + 64];
      }
      const int src2 = item_prev[r_];
      int src2_k = -1;
      if (src2 >= 0) src2_k = kjump[t_of_row[src2]];
      float lx = 0.0f;
      if constexpr (C::MASKED) {
        // gradient w.r.t. the input x of the segment that FOLLOWS this row (its prediction y):
        // W1x^T applied to the accumulated delta1, times tanh'
        float R[4];
        dpp_replicate(d1acc, R);
        const float din = chain_dot_lds<W>(tX1, R, 0.0f);
        const float tv = tanh_f(y);
        lx = uD ? din * (1.0f - tv * tv) : 0.0f;
      }
      // gradient of compute_loss at this row
      float dy, dybj;
      {
        const float e = x - y;
        const float f = a.loss_easy ? (ybj - x) : (ybj - y);
        const float sa = chain_sum<D>(m * e * e);
        const float sb = chain_sum<D>(m * f * f);
        const float scale = a.inv_batch * __builtin_amdgcn_rcpf((float)a.n_obs_ot[b]);
        const float na = sqrtf(sa + 1e-10f), nb = sqrtf(sb + 1e-10f);
        const float ca = a.loss_easy ? a.weight : 2.0f * a.weight;
        const float cb = a.loss_easy ? (1.0f - a.weight) : 2.0f * (1.0f - a.weight);
        const float s = ca * na + cb * nb;
        const float gg = 2.0f * s * scale;
        const float ga = gg * ca / na, gb = gg * cb / nb;
        if (a.loss_easy) {
          dy = -ga * m * e;
          dybj = gb * m * f;
        } else {
          dy = -ga * m * e - gb * m * f;
          dybj = gb * m * f;
        }
      }
      if constexpr (C::MASKED) dy += lx;   // last_X <- Y (models.py:483-484)
      if (uO) a.g_y[(size_t)r_ * DO + u] = dy;
      const float dh = dec_adj(hn, dy, ja[2][0], ja[2][1]);
      const float lam_hn = lam + dh;
      if (uH) a.g_hnew[(size_t)r_ * H + u] = lam_hn;
      if constexpr (C::MASKED) {
        // h_new = encoder(x_in, M), x_in = X M + (1 - M) y_bj (models.py:465-469)
        const float txin = tanh_f(x * m + (1.0f - m) * ybj);
        float R[4];
        dpp_replicate(lam_hn, R);
        float g = chain_dot_lds<H>(tE3, R, 0.0f);
        dpp_replicate(chain_delta<C::ACT, DROP>(g, ja[1][1], inv_keep, keepf), R);
        g = chain_dot_lds<W>(tE2, R, 0.0f);
        dpp_replicate(chain_delta<C::ACT, DROP>(g, ja[1][0], inv_keep, keepf), R);
        const float din = chain_dot_lds<W>(tE1, R, 0.0f);
        float v = din * (1.0f - txin * txin);
        if constexpr (C::ENC_CASE == 1) v += lam_hn;
        dybj += uD ? v * (1.0f - m) : 0.0f;
      }
      if (uO) a.g_ybj[(size_t)r_ * DO + u] = dybj;
      lam = dec_adj(hp, dybj, ja[0][0], ja[0][1]);
      h_after = hp;   // (should another jump sit in front of the same step)
      d1acc = 0.0f;
      src = chain_sgpr(src2);   // (the loop-carried bookkeeping stays in scalar registers)
      src_k = chain_sgpr(src2_k);
    }
  }
  if (uH) a.g_hstart[(size_t)b * H + u] = lam;
}

}  // namespace njode
