// njode_kernels.h -- the gfx950 kernels of the NJ-ODE hot path, templated on the
// model configuration.  Reference semantics: NJODE/models.py:379-518 (forward),
// :71-126 (loss), :188-199 (ODE rhs), :261-276 (encoder / readout).
//
// Two execution plans:
//   segment plan  : k_encode_rows -> k_ode_fwd_items -> k_jump_rows
//                   [-> k_jump_rows_bwd -> k_ode_bwd_items -> k_encode_rows_bwd]
//   lockstep plan : k_paths_fwd
#pragma once
#include "njode_device.h"

namespace njode {

constexpr uint32_t NET_ODE = 0, NET_ENC = 1, NET_DEC = 2, NET_DEC_BJ = 3, NET_DEC_ROW = 4;
constexpr uint32_t TKEY_START = 0xffffffffu;

template <int D_, int H_, int DO_, int NH_, int W_, int ACT_, bool MASKED_, bool CURT_,
          bool RES_, bool RNN_ = false>
struct Cfg {
  static constexpr int D = D_, H = H_, DO = DO_, NH = NH_, W = (NH_ > 0 ? W_ : 1), ACT = ACT_;
  static constexpr bool MASKED = MASKED_, CURT = CURT_, RES = RES_, RNN = RNN_;
  static constexpr int ODE_IN = D + H + (CURT ? 3 : 2);
  static constexpr int ENC_IN = MASKED ? 2 * D : D;
  using Ode = NetL<ODE_IN, H, NH, W_>;
  using Enc = NetL<ENC_IN, H, NH, W_>;
  using Dec = NetL<H, DO, NH, W_>;
  static constexpr int OFF_ODE = 0, OFF_ENC = Ode::SIZE, OFF_DEC = Ode::SIZE + Enc::SIZE;
  // use_rnn: GRU jump (models.py:202-217): nn.GRUCell parameters after the readout, in
  // state_dict order weight_ih [3H][D], weight_hh [3H][H], bias_ih [3H], bias_hh [3H]
  static constexpr int OFF_GRU = OFF_DEC + Dec::SIZE;
  static constexpr int G_WIH = 0, G_WHH = 3 * H * D, G_BIH = G_WHH + 3 * H * H, G_BHH = G_BIH + 3 * H;
  static constexpr int GRU_SIZE = G_BHH + 3 * H;
  static constexpr int P = OFF_GRU + (RNN ? GRU_SIZE : 0);
  static_assert(!(RNN && MASKED), "the GRU jump is not defined for masked models");
  // residual cases of FFNN (models.py:240-259)
  static constexpr int ENC_CASE = !RES ? 0 : (D <= H ? 1 : 2);
  static constexpr int DEC_CASE = !RES ? 0 : (H <= DO ? 1 : 2);
  static_assert(!RES || (D <= H ? H % D == 0 : D % H == 0), "residual: sizes must divide");
  static_assert(!RES || (H <= DO ? DO % H == 0 : H % DO == 0), "residual: sizes must divide");
  static_assert(!MASKED || D == DO, "masked mode imputes X with the readout");
};

struct KArgs {
  // parameters: flat vector and its transposed copy (weights stored [in][out])
  const float* P;
  const float* PT;
  float* frag;      // MFMA A-fragments of the ODE network (k_pack_frags)
  float* frag_enc;  // ... of the encoder and the readout (k_pack_net)
  float* frag_dec;
  float* frag2;     // scaled A-fragments of the ODE network (k_pack_frags2, njode_ode2.h)
  const void* fragx;  // split-bf16 A-fragments of the ODE network (prototype kernels of tools/ubench/njode_odex.h only)
  // segment plan: the part of the plan the encoder rows do not need is built on a helper
  // stream beside them; the ODE kernel waits for this event (null: everything on one stream)
  void* plan_ready;
  // host side only: the PlanJob (njode_plan.h) the ODE forward's launch carries in front of its own
  // blocks -- the NEXT batch's plan, deferred to this call (njode_plan_f32 with NJODE_C_PLAN_DEFER)
  const void* plan_job;
  // batch
  int B, n_obs;
  const float* start_X;
  const float* X;
  const float* M;
  const int* obs_idx;
  const int* n_obs_ot;
  float inv_batch;
  unsigned long long gid0;
  // schedule (device copies)
  int K, n_times;
  const float* step_dt;
  const float* step_t;
  const int* k_jump;
  const float* time_f32;
  // plan
  const int* t_of_row;
  const int* row_by_path;
  const int* path_sorted;
  const int* first_j;
  const int* first_row;
  const int* last_row;
  const int* item_prev;
  const int* item_next;
  const int* item_kbeg;
  const int* item_len;
  const int* order;
  const int* t_order;  // paths sorted by tail length (descending): hT items
  const long long* base_s;
  // stored activations of the ODE network (segment plan, matrix cores): block of the tile
  // [16 t, 16 t + 16) at Euler step s starts at chain record base16_s[s] + 16 t, where
  // base16_s = prefix sum of the per-step chain counts rounded up to 16
  const long long* base16_s;
  float* act;
  uint32_t* dbits;   // keep bits of the forward's four-wave role: [tile-step][lane], k1 | k2 << 16
  int dbits_ready;   // ... were written ahead of the ODE forward (by the fragment-pack launch)
  // masked lockstep forward (njode_mfma_lock4.h): dbits = [Euler step][tile][lane] words of the ODE
  // network, dbits_row = [row][3 evaluations][4 lane groups] words of the jumps' networks
  uint32_t* dbits_row;
  // intermediates
  float* h0row;
  float* h0start;
  float* h_end;
  float* lam_end;
  float* g_h0;
  float* lam_start;
  float* traj;
  // lockstep plan, saved for its backward: state before every Euler step (and the final
  // state) [K+1][B][H]; adjoint after every step [K][B][H]; source row of every step
  // [K][B]; per observation row: readout before / after the jump, their gradients, the
  // adjoint of the post-jump state; adjoint of every path's start state
  float* ltraj;
  float* lam_traj;
  int* src_row;
  float* y_row;
  float* ybj_row;
  float* g_y;
  float* g_ybj;
  float* g_hnew;
  float* g_hstart;
  // ... (masked models, njode_mfma_lock4.h) hidden activations of the ODE network at every step:
  // [K][tiles of 16 paths][4 waves][2 layers x 4 registers][64 lanes]
  float* lact;
  // ... and of the three network evaluations of every jump (readout before, encoder, readout
  // after): [row][4 waves][3 evaluations][2 layers][4 registers][4 lane groups]
  float* jact;
  float* loss_terms;
  float* slab;
  float* trash;  // [64 * max(H, D)] scratch target for the stores of inactive lanes
  // tile queue of the mixed ODE backward (njode_ode2.h): [0] four-wave tiles, [1] bulk tiles,
  // [2] finished blocks; tile_q_on: this launch pops its tiles (else static snake rounds)
  int* tile_q;
  int tile_q_on;
  int n_waves;      // persistent gradient kernels (VALU): waves == slab rows
  int n_waves_ode;  // same for the ODE backward kernel
  int n_waves_rows; // same for the row backward kernels on the matrix cores
  // lockstep backward: upstream gradient of hT [B][H] (NjodeBatch.grad_hT), or null
  const float* g_hT;
  // outputs
  float* hT;
  float* path_h;
  float* path_y;
  // options
  float weight;
  int loss_easy;
  int want_path, want_loss;  // lockstep plan: return_path / get_loss (wave-uniform)
  int save_traj;             // checkpoint states for the backward pass
  int defer_loss;            // segment plan: the backward's row kernel writes the loss terms (1: it runs in
                             // the backward call -- fused step; 2: in the forward call, NJODE_C_ROWS_IN_FWD)
  int ode_split;             // segment plan: ODE kernels with four waves per tile (njode_mfma_split.h)
  // ... mixed kernels: the first n_split_* blocks run the longest tiles four waves per tile,
  // the other blocks one wave per tile (njode_mfma_split.h); grid sizes
  int n_split_blocks, n_blocks_bwd, n_split_fwd, n_blocks_fwd;
  int q4_pt;       // masked lockstep kernels (njode_mfma_lock4.h): paths per 16-lane tile
  // masked lockstep kernels in the latency regime (njode_chain.h): one wave per path; lact / jact /
  // dbits / dbits_row then have that file's layouts
  int chain;
  // segment plan in the latency regime (njode_chain_seg.h): one wave per item; the ODE weight
  // gradients then come from the lockstep plan's (step, path) pair kernel
  int seg_chain;
  // ... and the deltas of the ODE network's two hidden layers at every (step, path) pair, stored by the
  // wave-per-chain sweeps beside the adjoint ([(k B + b)][delta1 | delta2][64 lanes]; null: the pair
  // kernel recomputes them)
  float* cdelta;
  // ... and per SEGMENT (lockstep plan: the steps behind row r -> record r, behind a path's start ->
  // record n_obs + b; segment plan: item r -> record r) S0 = sum delta1 | S1 = sum delta1 (t - tau) over its
  // steps, [64 lanes] each: the x / tau / time columns of dW1 (njode_chain_dw.h); grid of that kernel
  float* cseg;
  // segment plan on the matrix cores: the saving forward packs what the backward needs of every item
  // (Item::store_pack, [position in the item order][PACKF]) and, per tile of 16 positions, the record
  // base of the tile's LAST Euler step (base16_s[nmax - 1]); null: the backward walks the plan's arrays
  float* item_pack;
  long long* tile_last;
  int dw_pair_blocks, dw_seg_blocks;
  int dw_enc_fused;   // segment plan: k_encode_rows_bwd_mfma rides in that launch (njode_chain_dw.h)
  // the barrier counters (8 words) of the plan job this call's ODE forward hosts, when the fragment-pack
  // launch in front of it zeroes them (same stream: no memset launch of their own); else null
  unsigned* plan_sync_zero;
  // segment plan, round 5 (NJODE_ENC_FUSED=1): the one-wave role of k_ode_fwd_mixed evaluates
  // encoder(X) of an item's START row itself (njode_ode2.h); k_encode_rows_items covers the rest
  int enc_fused;
  DropCtx dc;
  float keep;
};

// Shapes the wave-per-path lockstep kernels (njode_chain.h) are written for, and their records
template <class C, bool TWO = (C::NH == 2)> struct ChainOk { static constexpr bool value = false; };
template <class C> struct ChainOk<C, true> {
  static constexpr bool value =
      C::MASKED && !C::RNN && C::W <= 64 && C::H <= 64 && C::D <= 64 && C::DO <= 64 && C::D == C::DO &&
      (C::ENC_CASE == 0 || (C::ENC_CASE == 1 && C::D == C::H)) &&
      (C::DEC_CASE == 0 || (C::DEC_CASE == 1 && C::DO == C::H));
};
template <class C, bool TWO = (C::NH == 2)> struct SegChainOk { static constexpr bool value = false; };
template <class C> struct SegChainOk<C, true> {
  static constexpr bool value = !C::MASKED && !C::RNN && C::W <= 64 && C::W > 16 && C::H <= 16 && C::D <= 8;
};
constexpr int CHAIN_MAX_WAVES = 8;              // waves (= paths) per block
constexpr int CHAIN_ACT_FLOATS = 2 * 64;        // stored hidden activations per (path, Euler step): [layer][lane]
constexpr int CHAIN_JACT_FLOATS = 3 * 2 * 64;   // ... per observation row: [evaluation][layer][lane]

// ---- small per-lane helpers -------------------------------------------------------
template <int N> NJ_DEV void load_vec(const float* p, float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = p[i];
}
template <int N> NJ_DEV void store_vec(float* p, const float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) p[i] = v[i];
}
NJ_DEV int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

template <class C, bool DROP> struct Masks {
  uint64_t m1 = 0, m2 = 0;
  NJ_DEV void draw(const KArgs& a, unsigned long long gid, uint32_t tkey, uint32_t net) {
    if constexpr (DROP && C::NH > 0) {
      uint32_t s = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32), tkey, net);
      m1 = keep_mask<C::W>(s, a.dc.thr16);
      if constexpr (C::NH > 1) m2 = keep_mask<C::W>(s, a.dc.thr16);
    }
  }
};

// encoder_map (FFNN, models.py:261-276): h = ffnn([tanh(x), mask]) (+ identity)
template <class C, bool DROP, class WP>
NJ_DEV void encode(WP Pe, const float (&x)[C::D], const float (&mask)[C::D],
                   float (&ein)[C::ENC_IN], float (&a1)[C::W], float (&a2)[C::W],
                   const Masks<C, DROP>& mk, float inv_keep, float (&h)[C::H]) {
#pragma unroll
  for (int i = 0; i < C::D; ++i) ein[i] = tanh_f(x[i]);
  if constexpr (C::MASKED) {
#pragma unroll
    for (int i = 0; i < C::D; ++i) ein[C::D + i] = mask[i];
  }
  net_fwd<typename C::Enc, C::ACT, DROP>(Pe, ein, h, a1, a2, mk.m1, mk.m2, inv_keep);
  if constexpr (C::ENC_CASE == 1) {
#pragma unroll
    for (int j = 0; j < C::H; ++j) h[j] += x[j % C::D];
  } else if constexpr (C::ENC_CASE == 2) {
    constexpr int mult = C::D / C::H;
#pragma unroll
    for (int j = 0; j < C::H; ++j) {
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < mult; ++c) s += x[c * C::H + j];
      h[j] += s * (1.0f / mult);
    }
  }
}

// readout_map: y = ffnn(tanh(h)) (+ identity); th receives tanh(h)
template <class C, bool DROP, class WP>
NJ_DEV void readout(WP Pd, const float (&h)[C::H], float (&th)[C::H], float (&a1)[C::W],
                    float (&a2)[C::W], const Masks<C, DROP>& mk, float inv_keep,
                    float (&y)[C::DO]) {
#pragma unroll
  for (int i = 0; i < C::H; ++i) th[i] = tanh_f(h[i]);
  net_fwd<typename C::Dec, C::ACT, DROP>(Pd, th, y, a1, a2, mk.m1, mk.m2, inv_keep);
  if constexpr (C::DEC_CASE == 1) {
#pragma unroll
    for (int j = 0; j < C::DO; ++j) y[j] += h[j % C::H];
  } else if constexpr (C::DEC_CASE == 2) {
    constexpr int mult = C::H / C::DO;
#pragma unroll
    for (int j = 0; j < C::DO; ++j) {
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < mult; ++c) s += h[c * C::DO + j];
      y[j] += s * (1.0f / mult);
    }
  }
}

NJ_DEV float sigmoid_f(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// GRU jump (nn.GRUCell on tanh(X_obs), tanh(h)): gates kept for the backward
template <class C>
NJ_DEV void gru_jump(cfp Pg, const float (&x)[C::D], const float (&h)[C::H], float (&tx)[C::D],
                     float (&th)[C::H], float (&r)[C::H], float (&z)[C::H], float (&n)[C::H],
                     float (&ghn)[C::H], float (&hn)[C::H]) {
  constexpr int H = C::H, D = C::D;
  float gi[3 * H], gh[3 * H];
#pragma unroll
  for (int i = 0; i < D; ++i) tx[i] = tanh_f(x[i]);
#pragma unroll
  for (int i = 0; i < H; ++i) th[i] = tanh_f(h[i]);
  const cfp P = launder(Pg);
  dense<D, 3 * H>(P + C::G_WIH, P + C::G_BIH, tx, gi);
  dense<H, 3 * H>(P + C::G_WHH, P + C::G_BHH, th, gh);
#pragma unroll
  for (int i = 0; i < H; ++i) {
    r[i] = sigmoid_f(gi[i] + gh[i]);
    z[i] = sigmoid_f(gi[H + i] + gh[H + i]);
    ghn[i] = gh[2 * H + i];
    n[i] = tanh_f(gi[2 * H + i] + r[i] * ghn[i]);
    hn[i] = (1.0f - z[i]) * n[i] + z[i] * th[i];
  }
  pin(hn);
}

// ODE input vector (models.py:188-199)
template <class C>
NJ_DEV void ode_input(const float (&tx)[C::D], const float (&h)[C::H], float tau, float t,
                      float (&in0)[C::ODE_IN]) {
#pragma unroll
  for (int i = 0; i < C::D; ++i) in0[i] = tx[i];
#pragma unroll
  for (int i = 0; i < C::H; ++i) in0[C::D + i] = tanh_f(h[i]);
  const float tdiff = t - tau;
  in0[C::D + C::H] = tau;
  in0[C::D + C::H + 1] = tdiff;
  if constexpr (C::CURT) in0[C::D + C::H + 2] = tau + tdiff;
}

// paper loss of one observation row and its gradients (models.py:71-126)
template <class C>
NJ_DEV float loss_row(const float (&x)[C::D], const float (&mask)[C::D], const float (&y)[C::DO],
                      const float (&ybj)[C::DO], float w, int easy, float scale,
                      float (&dy)[C::DO], float (&dybj)[C::DO]) {
  static_assert(C::D == C::DO, "loss compares X with the readout");
  float sa = 0.0f, sb = 0.0f;
#pragma unroll
  for (int q = 0; q < C::D; ++q) {
    const float m = C::MASKED ? mask[q] : 1.0f;
    const float e = x[q] - y[q];
    const float f = easy ? (ybj[q] - x[q]) : (ybj[q] - y[q]);
    sa = fmaf(m * e, e, sa);
    sb = fmaf(m * f, f, sb);
  }
  const float na = sqrtf(sa + 1e-10f), nb = sqrtf(sb + 1e-10f);
  const float ca = easy ? w : 2.0f * w, cb = easy ? (1.0f - w) : 2.0f * (1.0f - w);
  const float s = ca * na + cb * nb;
  const float g = 2.0f * s * scale;
  const float ga = g * ca / na, gb = g * cb / nb;
#pragma unroll
  for (int q = 0; q < C::D; ++q) {
    const float m = C::MASKED ? mask[q] : 1.0f;
    const float e = x[q] - y[q];
    if (easy) {
      const float f = ybj[q] - x[q];
      dy[q] = -ga * m * e;
      dybj[q] = gb * m * f;
    } else {
      const float f = ybj[q] - y[q];
      dy[q] = -ga * m * e - gb * m * f;
      dybj[q] = gb * m * f;
    }
  }
  return s * s * scale;
}

// =====================================================================================
// Segment plan
// =====================================================================================

// E: h0 of every segment: encoder on every observation row and every start value.
template <class C, bool DROP>
__global__ void __launch_bounds__(64) k_encode_rows(KArgs a) {
  const int tid = blockIdx.x * 64 + threadIdx.x;
  const int total = a.n_obs + a.B;
  if (tid >= total) return;
  const bool is_row = tid < a.n_obs;
  const int b = is_row ? a.obs_idx[tid] : tid - a.n_obs;
  const float* xp = is_row ? a.X + (size_t)tid * C::D : a.start_X + (size_t)b * C::D;
  float x[C::D], mask[C::D], ein[C::ENC_IN], a1[C::W], a2[C::W], h[C::H];
  load_vec(xp, x);
#pragma unroll
  for (int i = 0; i < C::D; ++i) mask[i] = 0.0f;
  Masks<C, DROP> mk;
  mk.draw(a, a.gid0 + b, is_row ? (uint32_t)a.k_jump[a.t_of_row[tid]] : TKEY_START, NET_ENC);
  encode<C, DROP>(as_cfp(a.P) + C::OFF_ENC, x, mask, ein, a1, a2, mk, a.dc.inv_keep, h);
  store_vec(is_row ? a.h0row + (size_t)tid * C::H : a.h0start + (size_t)b * C::H, h);
}

// Per-item descriptor loaded by the ODE kernels.  Segment items (one per observation
// row, sorted by length, descending) run from the previous observation of their path to
// their own; tail items (TAIL, one per path) run from the path's last observation to
// the end of the schedule and only produce hT.
template <class C> struct Item {
  int r, b, n, kbeg, prev;
  float tau;
  float tx[C::D];
  template <bool TAIL> NJ_DEV void load(const KArgs& a, int j, bool valid) {
    const int jj = valid ? j : 0;
    if constexpr (TAIL) {
      b = r = a.t_order[jj];
      prev = a.last_row[b];
      kbeg = prev >= 0 ? a.k_jump[a.t_of_row[prev >= 0 ? prev : 0]] : 0;
      n = valid ? a.K - kbeg : 0;
    } else {
      r = a.order[jj];
      b = a.obs_idx[r];
      n = valid ? a.item_len[r] : 0;
      kbeg = a.item_kbeg[r];
      prev = a.item_prev[r];
    }
    const int pv = prev >= 0 ? prev : 0;
    const float tprev = a.n_obs > 0 ? a.time_f32[a.t_of_row[pv]] : 0.0f;
    tau = prev >= 0 ? tprev : 0.0f;
    const float* xp = prev >= 0 ? a.X + (size_t)pv * C::D : a.start_X + (size_t)b * C::D;
#pragma unroll
    for (int i = 0; i < C::D; ++i) tx[i] = tanh_f(xp[i]);
  }
  NJ_DEV const float* h0(const KArgs& a) const {
    return prev >= 0 ? a.h0row + (size_t)prev * C::H : a.h0start + (size_t)b * C::H;
  }
  // Round 6: what the ODE backward needs of the item, packed by the saving forward per position j of the
  // item order -- [r, n, kbeg, b | tau, tanh(x)[D], pad] -- so that the backward's tile prologue is ONE
  // load deep (order -> item arrays -> t_of_row -> time / X were four dependent loads in front of every
  // tile: 3.5 us, profiles/r05_bwd_fixed_costs.txt).  KArgs::item_pack; null: load() as before.
  static constexpr int PACKF = 4 + 4 * ((1 + C::D + 3) / 4);
  NJ_DEV void store_pack(float* pack, int j) const {
    typedef float v4 __attribute__((ext_vector_type(4)));
    float* p = pack + (size_t)j * PACKF;
    const v4 head = {__int_as_float(r), __int_as_float(n), __int_as_float(kbeg), __int_as_float(b)};
    *(v4*)p = head;
#pragma unroll
    for (int q = 0; q < (PACKF - 4) / 4; ++q) {
      v4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * q + e;   // 0: tau; 1 .. D: tanh(x)
        v[e] = i == 0 ? tau : (i <= C::D ? tx[i - 1 < C::D ? (i > 0 ? i - 1 : 0) : 0] : 0.0f);
      }
      *(v4*)(p + 4 + 4 * q) = v;
    }
  }
  NJ_DEV void load_pack(const float* pack, int j, bool valid) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const float* p = pack + (size_t)j * PACKF;
    const v4 head = *(const v4*)p;
    r = __float_as_int(head[0]);
    n = valid ? __float_as_int(head[1]) : 0;
    kbeg = __float_as_int(head[2]);
    b = __float_as_int(head[3]);
    prev = -1;   // (not packed: the backward does not read it)
#pragma unroll
    for (int q = 0; q < (PACKF - 4) / 4; ++q) {
      const v4 v = *(const v4*)(p + 4 + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * q + e;
        if (i == 0) tau = v[e];
        else if (i <= C::D) tx[i - 1 < C::D ? i - 1 : 0] = v[e];
      }
    }
  }
};

// B: Euler evolve of every item from its h0 to the state just before its jump (or, for
// tail items, to the end of the schedule).  WLDS: the ODE network's weights are staged
// in LDS once per 256-thread block and read with wave-uniform ds_reads; otherwise they
// come through the scalar cache (s_load).
template <class C, bool DROP, bool TAIL, bool WLDS>
__global__ void __launch_bounds__(WLDS ? 256 : 64) k_ode_fwd_items(KArgs a) {
  const bool SAVE = !TAIL && a.save_traj != 0;  // wave-uniform
  constexpr int NT = WLDS ? 256 : 64;
  using NL = typename C::Ode;
  __shared__ __attribute__((aligned(16))) float wl[WLDS ? NL::SIZE : 4];
  if constexpr (WLDS) {
    for (int i = threadIdx.x; i < NL::SIZE; i += NT) wl[i] = a.P[C::OFF_ODE + i];
    __syncthreads();
  }
  const int n_items = TAIL ? a.B : a.n_obs;
  const int j = blockIdx.x * NT + threadIdx.x;
  const bool valid = j < n_items;
  Item<C> it;
  it.template load<TAIL>(a, j, valid);
  float h[C::H];
  load_vec(it.h0(a), h);
  const int nmax = wave_max(it.n);
  float* const trash = a.trash + threadIdx.x * C::H;
  for (int s = 0; s < nmax; ++s) {
    const bool active = s < it.n;
    const int k = active ? it.kbeg + s : 0;
    if (SAVE) {
      float* dst = active ? a.traj + (size_t)(a.base_s[s] + j) * C::H : trash;
      store_vec(dst, h);
    }
    const float dt = active ? a.step_dt[k] : 0.0f, t = a.step_t[k];
    float in0[C::ODE_IN], a1[C::W], a2[C::W], f[C::H];
    ode_input<C>(it.tx, h, it.tau, t, in0);
    Masks<C, DROP> mk;
    mk.draw(a, a.gid0 + it.b, (uint32_t)k, NET_ODE);
    if constexpr (WLDS)
      net_fwd<NL, C::ACT, DROP>((lcp)wl, in0, f, a1, a2, mk.m1, mk.m2, a.dc.inv_keep);
    else
      net_fwd<NL, C::ACT, DROP>(as_cfp(a.P) + C::OFF_ODE, in0, f, a1, a2, mk.m1, mk.m2,
                                a.dc.inv_keep);
#pragma unroll
    for (int i = 0; i < C::H; ++i) h[i] = fmaf(dt, f[i], h[i]);  // dt == 0 when inactive
  }
  float* out = TAIL ? a.hT + (size_t)it.b * C::H : a.h_end + (size_t)it.r * C::H;
  store_vec(valid ? out : trash, h);
}

// A (forward): readout before and after the jump, loss term of each row.
template <class C, bool DROP>
__global__ void __launch_bounds__(64) k_jump_rows(KArgs a) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= a.n_obs) return;
  const int b = a.obs_idx[r];
  const uint32_t tkey = (uint32_t)a.k_jump[a.t_of_row[r]];
  const cfp Pd = as_cfp(a.P) + C::OFF_DEC;
  float h[C::H], th[C::H], a1[C::W], a2[C::W], y[C::DO], ybj[C::DO], x[C::D], mask[C::D];
  Masks<C, DROP> mk;
  load_vec(a.h_end + (size_t)r * C::H, h);
  mk.draw(a, a.gid0 + b, tkey, NET_DEC_BJ);
  readout<C, DROP>(Pd, h, th, a1, a2, mk, a.dc.inv_keep, ybj);
  load_vec(a.h0row + (size_t)r * C::H, h);
  mk.draw(a, a.gid0 + b, tkey, NET_DEC);
  readout<C, DROP>(launder(Pd), h, th, a1, a2, mk, a.dc.inv_keep, y);
  load_vec(a.X + (size_t)r * C::D, x);
#pragma unroll
  for (int i = 0; i < C::D; ++i) mask[i] = 1.0f;
  float dy[C::DO], dybj[C::DO];
  const float scale = a.inv_batch / (float)a.n_obs_ot[b];
  a.loss_terms[r] = loss_row<C>(x, mask, y, ybj, a.weight, a.loss_easy, scale, dy, dybj);
}

// Gradient of the readout's identity path + tanh at its input
template <class C>
NJ_DEV void readout_input_grad(const float (&din)[C::H], const float (&th)[C::H],
                               const float (&dy)[C::DO], float (&dh)[C::H]) {
#pragma unroll
  for (int i = 0; i < C::H; ++i) dh[i] = din[i] * (1.0f - th[i] * th[i]);
  if constexpr (C::DEC_CASE == 1) {
#pragma unroll
    for (int j = 0; j < C::DO; ++j) dh[j % C::H] += dy[j];
  } else if constexpr (C::DEC_CASE == 2) {
    constexpr int mult = C::H / C::DO;
#pragma unroll
    for (int i = 0; i < C::H; ++i) dh[i] += dy[i % C::DO] * (1.0f / mult);
  }
}

// A (backward): d loss / d readout params; adjoint at the segment end (lam_end) and
// the loss' direct gradient on the post-jump state (g_h0).
template <class C, bool DROP>
__global__ void __launch_bounds__(64, 2) k_jump_rows_bwd(KArgs a) {
  using NL = typename C::Dec;
  __shared__ float lds_raw[NetAcc<NL>::LDS_FLOATS];
  lfp lds = (lfp)lds_raw;
  const int lane = threadIdx.x;
  const int wave = blockIdx.x;
  NetAcc<NL> g;
  g.zero();
  const cfp Pd0 = as_cfp(a.P) + C::OFF_DEC, PTd0 = as_cfp(a.PT) + C::OFF_DEC;
  const int n_tiles = (a.n_obs + 63) / 64;
  for (int tile = wave; tile < n_tiles; tile += a.n_waves) {
    const int r0 = tile * 64 + lane;
    const bool valid = r0 < a.n_obs;
    const int r = valid ? r0 : 0;
    const int b = a.obs_idx[r];
    const unsigned long long gid = a.gid0 + b;
    const uint32_t tkey = (uint32_t)a.k_jump[a.t_of_row[r]];
    const cfp Pd = launder(Pd0), PTd = launder(PTd0);
    float h[C::H], th[C::H], a1[C::W], a2[C::W], y[C::DO], ybj[C::DO], x[C::D], mask[C::D];
    float dy[C::DO], dybj[C::DO], din[C::H], dh[C::H];
    Masks<C, DROP> mk_bj, mk;
    mk_bj.draw(a, gid, tkey, NET_DEC_BJ);
    mk.draw(a, gid, tkey, NET_DEC);
    // forward: y_bj first (activations discarded), then y (activations kept)
    load_vec(a.h_end + (size_t)r * C::H, h);
    readout<C, DROP>(Pd, h, th, a1, a2, mk_bj, a.dc.inv_keep, ybj);
    load_vec(a.h0row + (size_t)r * C::H, h);
    readout<C, DROP>(Pd, h, th, a1, a2, mk, a.dc.inv_keep, y);
    load_vec(a.X + (size_t)r * C::D, x);
#pragma unroll
    for (int i = 0; i < C::D; ++i) mask[i] = 1.0f;
    // branch-free on purpose: a divergent branch here splits the block and makes the
    // compiler keep a whole network's scalar-loaded weights live across it
    const float scale = (valid ? a.inv_batch : 0.0f) * __builtin_amdgcn_rcpf((float)a.n_obs_ot[b]);
    loss_row<C>(x, mask, y, ybj, a.weight, a.loss_easy, scale, dy, dybj);
    // backward through y = readout(h0row[r])
    net_bwd<NL, C::ACT, DROP, 0, C::H>(PTd, lds, g, th, dy, a1, a2, mk.m1, mk.m2,
                                       a.dc.inv_keep, a.keep, din, lane);
    readout_input_grad<C>(din, th, dy, dh);
    store_vec(valid ? a.g_h0 + (size_t)r * C::H : a.trash + lane * C::H, dh);
    // backward through y_bj = readout(h_end[r]) (recompute its activations)
    load_vec(a.h_end + (size_t)r * C::H, h);
    readout<C, DROP>(launder(Pd), h, th, a1, a2, mk_bj, a.dc.inv_keep, ybj);
    net_bwd<NL, C::ACT, DROP, 0, C::H>(launder(PTd), lds, g, th, dybj, a1, a2, mk_bj.m1,
                                       mk_bj.m2, a.dc.inv_keep, a.keep, din, lane);
    readout_input_grad<C>(din, th, dybj, dh);
    store_vec(valid ? a.lam_end + (size_t)r * C::H : a.trash + lane * C::H, dh);
  }
  g.flush(a.slab + (size_t)wave * C::P + C::OFF_DEC, lane);
}

// C: reverse Euler sweep of every segment (exact discrete adjoint), d loss / d ODE params.
// WLDS: weights (W and its transposed copy) staged in LDS once per 256-thread block
// (4 waves, each with its own staging rows, 16 chains per phase).
template <class C, bool DROP, bool WLDS>
__global__ void __launch_bounds__(WLDS ? 256 : 64, 2) k_ode_bwd_items(KArgs a) {
  using NL = typename C::Ode;
  constexpr int CHN = WLDS ? 16 : 32;
  constexpr int NW = WLDS ? 4 : 1, NT = NW * 64;
  using Acc = NetAcc<NL, CHN>;
  __shared__ __attribute__((aligned(16))) float
      lds_raw[NW * Acc::LDS_FLOATS + (WLDS ? 2 * NL::SIZE : 0)];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * NW + wv;
  lfp lds = (lfp)lds_raw + wv * Acc::LDS_FLOATS;
  if constexpr (WLDS) {
    float* wl = lds_raw + NW * Acc::LDS_FLOATS;
    for (int i = threadIdx.x; i < NL::SIZE; i += NT) {
      wl[i] = a.P[C::OFF_ODE + i];
      wl[NL::SIZE + i] = a.PT[C::OFF_ODE + i];
    }
    __syncthreads();
  }
  const lcp Wl = (lcp)(lds_raw + NW * Acc::LDS_FLOATS), WTl = Wl + NL::SIZE;
  const cfp Ws = as_cfp(a.P) + C::OFF_ODE, WTs = as_cfp(a.PT) + C::OFF_ODE;
  Acc g;
  g.zero();
  float* const trash = a.trash + threadIdx.x * C::H;
  const int n_tiles = (a.n_obs + 63) / 64;
  for (int tile = wave; tile < n_tiles; tile += a.n_waves) {
    const int j = tile * 64 + lane;
    const bool valid = j < a.n_obs;
    Item<C> it;
    it.template load<false>(a, j, valid);
    float lam[C::H];
    load_vec(a.lam_end + (size_t)it.r * C::H, lam);  // r is clamped to a valid row
#pragma unroll
    for (int i = 0; i < C::H; ++i) lam[i] = valid ? lam[i] : 0.0f;
    const int nmax = wave_max(it.n);
    for (int s = nmax - 1; s >= 0; --s) {
      const bool active = s < it.n;
      const int k = active ? it.kbeg + s : 0;
      float h[C::H];
      // inactive lanes read slot 0 (always allocated when the loop runs); their
      // contribution is cancelled by dt = 0 below
      load_vec(a.traj + (active ? (size_t)(a.base_s[s] + j) * C::H : 0), h);
      const float dt = active ? a.step_dt[k] : 0.0f, t = a.step_t[k];
      float in0[C::ODE_IN], a1[C::W], a2[C::W], f[C::H], dout[C::H], din[C::H];
      ode_input<C>(it.tx, h, it.tau, t, in0);
      Masks<C, DROP> mk;
      mk.draw(a, a.gid0 + it.b, (uint32_t)k, NET_ODE);
      if constexpr (WLDS)
        net_fwd<NL, C::ACT, DROP>(Wl, in0, f, a1, a2, mk.m1, mk.m2, a.dc.inv_keep);
      else
        net_fwd<NL, C::ACT, DROP>(Ws, in0, f, a1, a2, mk.m1, mk.m2, a.dc.inv_keep);
      // h' = h + dt f(h): d/df = dt * lam (zero for inactive lanes since dt = 0)
#pragma unroll
      for (int i = 0; i < C::H; ++i) dout[i] = dt * lam[i];
      if constexpr (WLDS)
        net_bwd<NL, C::ACT, DROP, C::D, C::D + C::H>(WTl, lds, g, in0, dout, a1, a2, mk.m1,
                                                     mk.m2, a.dc.inv_keep, a.keep, din, lane);
      else
        net_bwd<NL, C::ACT, DROP, C::D, C::D + C::H>(WTs, lds, g, in0, dout, a1, a2, mk.m1,
                                                     mk.m2, a.dc.inv_keep, a.keep, din, lane);
#pragma unroll
      for (int i = 0; i < C::H; ++i) {
        const float th = in0[C::D + i];
        lam[i] = fmaf(din[i], 1.0f - th * th, lam[i]);
      }
    }
    store_vec(valid ? a.lam_start + (size_t)it.r * C::H : trash, lam);
  }
  g.flush(a.slab + (size_t)wave * C::P + C::OFF_ODE, lane);
}

// D: d loss / d encoder params from the adjoints at every segment start.
template <class C, bool DROP>
__global__ void __launch_bounds__(64, 2) k_encode_rows_bwd(KArgs a) {
  using NL = typename C::Enc;
  __shared__ float lds_raw[NetAcc<NL>::LDS_FLOATS];
  lfp lds = (lfp)lds_raw;
  const int lane = threadIdx.x;
  const int wave = blockIdx.x;
  NetAcc<NL> g;
  g.zero();
  const cfp Pe0 = as_cfp(a.P) + C::OFF_ENC, PTe0 = as_cfp(a.PT) + C::OFF_ENC;
  const int total = a.n_obs + a.B;
  const int n_tiles = (total + 63) / 64;
  for (int tile = wave; tile < n_tiles; tile += a.n_waves) {
    const int t0 = tile * 64 + lane;
    const bool valid = t0 < total;
    const int tid = valid ? t0 : 0;
    const bool is_row = tid < a.n_obs;
    const int b = is_row ? a.obs_idx[tid] : tid - a.n_obs;
    const float* xp = is_row ? a.X + (size_t)tid * C::D : a.start_X + (size_t)b * C::D;
    float x[C::D], mask[C::D], ein[C::ENC_IN], a1[C::W], a2[C::W], h[C::H], gh[C::H], din[1];
    load_vec(xp, x);
#pragma unroll
    for (int i = 0; i < C::D; ++i) mask[i] = 0.0f;
    {
      // adjoint of this start state: the next segment's sweep result, plus (for an
      // observation row) the loss' direct gradient through y = readout(h0).
      // Branch-free: clamped loads, then selects.
      const int nxt = is_row ? a.item_next[tid] : a.first_row[b];
      const int nx = nxt >= 0 ? nxt : 0;
      const int rr = is_row ? tid : 0;
#pragma unroll
      for (int i = 0; i < C::H; ++i) {
        const float l = a.lam_start[(size_t)nx * C::H + i];
        const float q = a.g_h0[(size_t)rr * C::H + i];
        gh[i] = valid ? ((nxt >= 0 ? l : 0.0f) + (is_row ? q : 0.0f)) : 0.0f;
      }
    }
    Masks<C, DROP> mk;
    mk.draw(a, a.gid0 + b, is_row ? (uint32_t)a.k_jump[a.t_of_row[tid]] : TKEY_START, NET_ENC);
    encode<C, DROP>(launder(Pe0), x, mask, ein, a1, a2, mk, a.dc.inv_keep, h);
    net_bwd<NL, C::ACT, DROP, 0, 0>(launder(PTe0), lds, g, ein, gh, a1, a2, mk.m1, mk.m2,
                                    a.dc.inv_keep, a.keep, din, lane);
  }
  g.flush(a.slab + (size_t)wave * C::P + C::OFF_ENC, lane);
}

// =====================================================================================
// Lockstep plan: one lane per path over the shared grid (all modes, forward only)
// =====================================================================================
template <class C, bool DROP>
__global__ void __launch_bounds__(64) k_paths_fwd(KArgs a) {
  const bool PATH = a.want_path != 0, LOSS = a.want_loss != 0;  // uniform branches
  const bool SAVE = a.save_traj != 0;
  const int b0 = blockIdx.x * 64 + threadIdx.x;
  const bool valid = b0 < a.B;
  const int b = valid ? b0 : a.B - 1;
  const unsigned long long gid = a.gid0 + b;
  const cfp Po0 = as_cfp(a.P) + C::OFF_ODE, Pe0 = as_cfp(a.P) + C::OFF_ENC,
            Pd0 = as_cfp(a.P) + C::OFF_DEC;
  const cfp kj = as_cfp((const float*)a.k_jump), sdt = as_cfp(a.step_dt),
            stt = as_cfp(a.step_t), tf = as_cfp(a.time_f32);
  const int __attribute__((address_space(4)))* kjump =
      (const int __attribute__((address_space(4)))*)kj;

  float xl[C::D], mask[C::D], h[C::H], y[C::DO] = {};
  float ein[C::ENC_IN], a1[C::W], a2[C::W], th[C::H];
  load_vec(a.start_X + (size_t)b * C::D, xl);
#pragma unroll
  for (int i = 0; i < C::D; ++i) mask[i] = 0.0f;
  Masks<C, DROP> mk;
  mk.draw(a, gid, TKEY_START, NET_ENC);
  encode<C, DROP>(Pe0, xl, mask, ein, a1, a2, mk, a.dc.inv_keep, h);
  float tx[C::D];
#pragma unroll
  for (int i = 0; i < C::D; ++i) tx[i] = tanh_f(xl[i]);
  float tau = 0.0f, loss_acc = 0.0f;

  int cur = a.first_j[b];
  int next_i = a.n_obs > 0 ? a.t_of_row[a.row_by_path[cur >= 0 ? cur : 0]] : 0;
  next_i = cur >= 0 ? next_i : 0x7fffffff;
  int src = -1;  // row of this path's most recent jump
  int row = 0;
  auto emit = [&](uint32_t tkey) {
    if (PATH) {
      mk.draw(a, gid, tkey, NET_DEC_ROW);
      readout<C, DROP>(launder(Pd0), h, th, a1, a2, mk, a.dc.inv_keep, y);
    }
  };
  // stores are branch-free (inactive lanes write to a scratch line): a divergent
  // `if` around the only use of a network's output makes the compiler sink the FMAs
  // into it and keep the whole network's scalar-loaded weights live (SGPR spills)
  auto write_row = [&]() {
    if (PATH) {
      store_vec(valid ? a.path_h + ((size_t)row * a.B + b) * C::H : a.trash + threadIdx.x * C::H, h);
      store_vec(valid ? a.path_y + ((size_t)row * a.B + b) * C::DO
                      : a.trash + threadIdx.x * C::DO, y);
      ++row;
    }
  };
  emit(TKEY_START - 1);
  write_row();

  int i = 0;
  for (int k = 0;; ++k) {
    while (i < a.n_times && kjump[i] == k) {
      const bool has = valid && next_i == i;
      if (__any(has)) {
        if (has) {
          const int r = a.row_by_path[cur];
          float ybj[C::DO], x[C::D], xin[C::D];
          if (SAVE) store_vec(a.h_end + (size_t)r * C::H, h);  // state before the jump
          mk.draw(a, gid, (uint32_t)k, NET_DEC_BJ);
          readout<C, DROP>(launder(Pd0), h, th, a1, a2, mk, a.dc.inv_keep, ybj);
          load_vec(a.X + (size_t)r * C::D, x);
          if constexpr (C::MASKED) {
            load_vec(a.M + (size_t)r * C::D, mask);
#pragma unroll
            for (int q = 0; q < C::D; ++q)
              xin[q] = x[q] * mask[q] + (1.0f - mask[q]) * ybj[q];
          } else {
#pragma unroll
            for (int q = 0; q < C::D; ++q) { xin[q] = x[q]; mask[q] = 1.0f; }
          }
          if constexpr (C::RNN) {
            float gtx[C::D], gth[C::H], gr[C::H], gz[C::H], gn[C::H], gg[C::H], hn[C::H];
            gru_jump<C>(as_cfp(a.P) + C::OFF_GRU, xin, h, gtx, gth, gr, gz, gn, gg, hn);
#pragma unroll
            for (int q = 0; q < C::H; ++q) h[q] = hn[q];
          } else {
            mk.draw(a, gid, (uint32_t)k, NET_ENC);
            encode<C, DROP>(launder(Pe0), xin, mask, ein, a1, a2, mk, a.dc.inv_keep, h);
          }
          mk.draw(a, gid, (uint32_t)k, NET_DEC);
          readout<C, DROP>(launder(Pd0), h, th, a1, a2, mk, a.dc.inv_keep, y);
          if (LOSS) {
            float dy[C::DO], dybj[C::DO];
            const float scale = a.inv_batch / (float)a.n_obs_ot[b];
            loss_acc += loss_row<C>(x, mask, y, ybj, a.weight, a.loss_easy, scale, dy, dybj);
          }
          if (SAVE) {
            store_vec(a.y_row + (size_t)r * C::DO, y);
            store_vec(a.ybj_row + (size_t)r * C::DO, ybj);
            src = r;
          }
          // last_X <- Y (masked) or X_obs; tau <- obs time (models.py:481-489)
#pragma unroll
          for (int q = 0; q < C::D; ++q) tx[q] = tanh_f(C::MASKED ? y[q] : x[q]);
          tau = tf[i];
          ++cur;
          {
            const int cc = cur < a.n_obs ? cur : 0;
            const int nt_ = a.t_of_row[a.row_by_path[cc]];
            next_i = (cur < a.n_obs && a.path_sorted[cc] == b) ? nt_ : 0x7fffffff;
          }
        }
      }
      write_row();
      ++i;
    }
    if (SAVE) {  // state before step k (after the jumps applied at k); k == K: final state
      store_vec(valid ? a.ltraj + ((size_t)k * a.B + b) * C::H : a.trash + threadIdx.x * C::H, h);
      if (valid && k < a.K) a.src_row[(size_t)k * a.B + b] = src;
    }
    if (k >= a.K) break;
    {
      const float dt = sdt[k], t = stt[k];
      float in0[C::ODE_IN], f[C::H];
      ode_input<C>(tx, h, tau, t, in0);
      mk.draw(a, gid, (uint32_t)k, NET_ODE);
      net_fwd<typename C::Ode, C::ACT, DROP>(launder(Po0), in0, f, a1, a2, mk.m1, mk.m2,
                                             a.dc.inv_keep);
#pragma unroll
      for (int q = 0; q < C::H; ++q) h[q] = fmaf(dt, f[q], h[q]);
      emit(0x80000000u + (uint32_t)k);
      write_row();
    }
  }
  store_vec(valid ? a.hT + (size_t)b * C::H : a.trash + threadIdx.x * C::H, h);
  if (LOSS && valid) a.loss_terms[b] = loss_acc;
}

}  // namespace njode
