// njode_cfg.hip -- one model shape, compiled once per entry of the build table
// (njode_amd/build.py) and per NJ_PART (0 segment forward + registration,
// 1 segment backward, 2 lockstep forward, 3 lockstep backward, 4 / 5 the wave-per-path lockstep
// forward / adjoint sweep of njode_chain.h) with
//   -DNJ_ID=.. -DNJ_D=.. -DNJ_H=.. -DNJ_DO=.. -DNJ_NH=.. -DNJ_W=.. -DNJ_ACT=..
//   -DNJ_MASKED=.. -DNJ_CURT=.. -DNJ_RES=.. -DNJ_PART=..
#include <cstdlib>
#include <cstring>

#include "njode_host.h"
#if NJ_PART >= 4
#include "njode_chain.h"
#include "njode_chain_seg.h"
#include "njode_chain_dw.h"
#endif

#define NJ_CAT_(a, b) a##b
#define NJ_CAT(a, b) NJ_CAT_(a, b)

namespace njode {

#ifndef NJ_RNN
#define NJ_RNN 0
#endif
using C = Cfg<NJ_D, NJ_H, NJ_DO, NJ_NH, NJ_W, NJ_ACT, (NJ_MASKED != 0), (NJ_CURT != 0),
              (NJ_RES != 0), (NJ_RNN != 0)>;

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

hipError_t NJ_CAT(njode_seg_forward_, NJ_ID)(const KArgs& a, bool drop, bool tails, int ode,
                                            hipStream_t st);
hipError_t NJ_CAT(njode_seg_backward_, NJ_ID)(const KArgs& a, bool drop, int ode, hipStream_t st);
constexpr bool HAS_MFMA = C::NH == 2 && !C::MASKED && !C::RNN && C::DO <= 16 && C::W < 64;
// the lockstep forward on the matrix cores also covers masked shapes
constexpr bool HAS_MFMA_LOCK = C::NH == 2 && !C::RNN && C::W < 64 && C::H <= 64 && C::DO <= 64;
// ... and so does its adjoint sweep, unless the encoder's identity path folds units
constexpr bool HAS_MFMA_SWEEP =
    HAS_MFMA_LOCK && (!C::MASKED || C::ENC_CASE == 0 || (C::ENC_CASE == 1 && C::D == C::H));
constexpr bool HAS_SPLIT = HAS_MFMA && SplitOk<C>::value;
// masked shapes: one tile over the four waves of a block (njode_mfma_lock4.h); NJODE_LOCK4=0
// keeps the one-wave kernels (maintainer A/B)
constexpr bool HAS_Q4 = HAS_MFMA_SWEEP && Q4Ok<C>::value;
// ... or, for small batches, one wave per path (njode_chain.h; the choice is KArgs::chain, made by
// njode_api.hip's make_layout)
constexpr bool HAS_CHAIN = HAS_Q4 && ChainOk<C>::value;
hipError_t NJ_CAT(njode_chain_forward_, NJ_ID)(const KArgs& a, bool drop, hipStream_t st);
hipError_t NJ_CAT(njode_chain_sweep_, NJ_ID)(const KArgs& a, bool drop, hipStream_t st);
// dW of the ODE network from the wave-per-chain sweeps' records (njode_chain_dw.h; part 5); false: not
// launched (no stored deltas / segment sums for this call) -- k_ode_dw_pairs_mfma then
bool NJ_CAT(njode_chain_dw_, NJ_ID)(const KArgs& a, hipStream_t st);
// ... the same with the segment plan's encoder pass (k_encode_rows_bwd_mfma) as a role of the launch
bool NJ_CAT(njode_chain_dw_enc_, NJ_ID)(const KArgs& a, bool drop, hipStream_t st);
// the segment plan's ODE kernels with one wave per item (njode_chain_seg.h; KArgs::seg_chain)
constexpr bool HAS_SEG_CHAIN = HAS_SPLIT && HAS_MFMA_SWEEP && SegChainOk<C>::value;
hipError_t NJ_CAT(njode_seg_chain_forward_, NJ_ID)(const KArgs& a, bool drop, bool tails, hipStream_t st);
hipError_t NJ_CAT(njode_seg_chain_backward_, NJ_ID)(const KArgs& a, bool drop, hipStream_t st);
static inline bool lock4_on() {
  static const bool on = [] {
    const char* e = getenv("NJODE_LOCK4");
    return !(e && e[0] == '0');
  }();
  return on;
}
template <bool ON, class CC> struct FragSize {
  static constexpr int ode = 0, enc = 0, dec = 0;
};
template <class CC> struct FragSize<true, CC> {
  static constexpr int ode = MF<CC>::NALL * 64;
  static constexpr int enc = EncS<CC>::type::NALL * 64;
  static constexpr int dec = DecS<CC>::type::NALL * 64;
};
using FS = FragSize<(HAS_MFMA || HAS_MFMA_LOCK), C>;
constexpr int MF_FLOATS = FS::ode + FS::enc + FS::dec + FS::ode;   // + the scaled ODE table (frag2)
constexpr int FRAG2_OFF = FS::ode + FS::enc + FS::dec;
template <bool ON, class CC> struct ActSize { static constexpr int value = 0; };
template <class CC> struct ActSize<true, CC> { static constexpr int value = StepRec<CC>::PER_CHAIN; };
constexpr int ACT_FLOATS = ActSize<HAS_SPLIT, C>::value;   // stored activations: demo-family shapes

// All fragment tables of the three networks in ONE launch instead of three in
// front of the encoder rows (each of these launches costs ~5 us of a chain that sits on the
// step's critical path once the plan is built ahead)
template <class C, class ES, class DS>
__global__ void k_pack_all(const float* __restrict__ P, float* __restrict__ frag, float* __restrict__ frag2,
                           float* __restrict__ frag_enc, float* __restrict__ frag_dec, float ik,
                           unsigned* __restrict__ zero8) {
  constexpr int N = MF<C>::NALL * 64, NE = ES::NALL * 64, ND = DS::NALL * 64;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (zero8 && idx < 8) zero8[idx] = 0;   // (KArgs::plan_sync_zero)
  if (idx < N) frag[idx] = ode_frag_value<C>(P, idx, 1.0f, 1.0f);
  else if (idx < 2 * N) frag2[idx - N] = ode_frag_value<C>(P, idx - N, Ode2Scale<C>::S, ik);
  else if (idx < 2 * N + NE) pack_net_value<typename C::Enc, ES>(P + C::OFF_ENC, frag_enc, idx - 2 * N);
  else if (idx < 2 * N + NE + ND) pack_net_value<typename C::Dec, DS>(P + C::OFF_DEC, frag_dec, idx - 2 * N - NE);
}
// ... and, in the same launch, the keep bits of the ODE forward's four-wave role (blocks behind the
// n_pack packing blocks: njode_mfma_split.h, drop_bits_tile_steps): no launch, no stream hop of
// their own, parallel over the chip, before the forward kernel starts
template <class C, class ES, class DS>
__global__ void k_pack_all_bits(KArgs a, int n_pack) {
  if ((int)blockIdx.x < n_pack) {
    constexpr int N = MF<C>::NALL * 64, NE = ES::NALL * 64, ND = DS::NALL * 64;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (a.plan_sync_zero && idx < 8) a.plan_sync_zero[idx] = 0;
    if (idx < N) a.frag[idx] = ode_frag_value<C>(a.P, idx, 1.0f, 1.0f);
    else if (idx < 2 * N) a.frag2[idx - N] = ode_frag_value<C>(a.P, idx - N, Ode2Scale<C>::S, a.dc.inv_keep);
    else if (idx < 2 * N + NE) pack_net_value<typename C::Enc, ES>(a.P + C::OFF_ENC, a.frag_enc, idx - 2 * N);
    else if (idx < 2 * N + NE + ND) pack_net_value<typename C::Dec, DS>(a.P + C::OFF_DEC, a.frag_dec, idx - 2 * N - NE);
  } else {
    if constexpr (SegChainOk<C>::value) {
      if (a.seg_chain) {   // (the wave-per-item forward's lane masks: njode_chain_seg.h)
        seg_chain_bits_body<C>(a, (int)blockIdx.x - n_pack, (int)gridDim.x - n_pack);
        return;
      }
    }
    if constexpr (HAS_SPLIT) {
      const int nb = (int)gridDim.x - n_pack;
      drop_bits_tile_steps<C>(a, ((int)blockIdx.x - n_pack) * 4 + (threadIdx.x >> 6), nb * 4);
    }
  }
}

// MFMA launches live in templates on the configuration so that `if constexpr` really
// discards them for shapes the matrix-core kernels are not written for
// (bits: the plan is complete on this stream -- the call has no helper stream -- and the forward
// will draw dropout masks: KArgs::dbits_ready)
template <class CC> static void launch_pack_frags(const KArgs& a, hipStream_t st, bool bits = false) {
  if constexpr (HAS_MFMA) {
    using ES = typename EncS<CC>::type;
    using DS = typename DecS<CC>::type;
    const int n_pack = cdiv((2 * MF<CC>::NALL + ES::NALL + DS::NALL) * 64, 256);
    if (bits) {
      // one wave per 8 Euler steps of a four-wide tile; the number of such tiles is known on the
      // device only: enough waves for the small plans (every tile four-wide), a persistent
      // grid for the large ones
      const long long work = a.seg_chain ? (long long)a.K * a.B / 64 + 1   // (256 (path, step) pairs per block)
                                         : (long long)cdiv(a.n_obs, 16) * cdiv(a.K > 0 ? a.K : 1, 8);
      const int nb = (int)(work / 4 + 1 < 1024 ? work / 4 + 1 : 1024);
      k_pack_all_bits<CC, ES, DS><<<n_pack + nb, 256, 0, st>>>(a, n_pack);
    } else {
      k_pack_all<CC, ES, DS><<<n_pack, 256, 0, st>>>(a.P, a.frag, a.frag2, a.frag_enc, a.frag_dec, a.dc.inv_keep,
                                                  a.plan_sync_zero);
    }
  }
}
template <class CC, bool DROP> static void launch_mfma_enc(const KArgs& a, hipStream_t st) {
  if constexpr (HAS_MFMA) {
    const int n_tiles = cdiv(a.n_obs + a.B, 16);
    // one-wave blocks, 64 VGPRs: four waves per SIMD hide the row gathers (obs_idx -> path,
    // t_of_row -> k_jump, X) each tile starts with: 51 -> 37 us for 219 470 rows (the step time
    // does not move: the plan on the helper stream is the critical path beside this kernel);
    // NJODE_ENC_BLOCKS overrides (A/B)
    static const int enc_blocks = getenv("NJODE_ENC_BLOCKS") ? atoi(getenv("NJODE_ENC_BLOCKS")) : 4096;
    k_encode_rows_mfma<CC, DROP><<<n_tiles < enc_blocks ? n_tiles : enc_blocks, 64, 0, st>>>(a);
  }
}
template <class CC, bool DROP> static void launch_mfma_jump(const KArgs& a, hipStream_t st) {
  if constexpr (HAS_MFMA) {
    const int n_tiles = cdiv(a.n_obs, 16);
    k_jump_rows_mfma<CC, DROP><<<n_tiles < 2048 ? n_tiles : 2048, 64, 0, st>>>(a);
  }
}
template <class CC, bool DROP> static void launch_ode_bwd_mfma(const KArgs& a, bool split, hipStream_t st) {
  if constexpr (HAS_MFMA) {
    if constexpr (HAS_SPLIT) {
      if (split) {
        ProfScope ps("k_ode_bwd_mixed", st);
        if (a.tile_q_on) k_ode_bwd_mixed<CC, DROP, true><<<a.n_blocks_bwd, 256, 0, st>>>(a);
        else k_ode_bwd_mixed<CC, DROP, false><<<a.n_blocks_bwd, 256, 0, st>>>(a);
        return;
      }
    }
    ProfScope ps("k_ode_bwd_mfma", st);
    k_ode_bwd_mfma<CC, DROP><<<a.n_waves_ode / 4, 256, 0, st>>>(a);
  }
}
template <class CC, bool DROP> static void launch_jump_rows_bwd(const KArgs& a, hipStream_t st) {
  if constexpr (HAS_MFMA) {
    ProfScope ps("k_jump_rows_bwd_mfma", st);
    k_jump_rows_bwd_mfma<CC, DROP><<<a.n_waves_rows / 4, 256, 0, st>>>(a);
  }
}
template <class CC, bool DROP>
static void launch_mfma_rows_bwd(const KArgs& a, bool split, hipStream_t st) {
  if constexpr (HAS_MFMA) {
    // (defer_loss == 2, NJODE_C_ROWS_IN_FWD: the forward call already ran this pass)
    if (a.defer_loss != 2) launch_jump_rows_bwd<CC, DROP>(a, st);
    bool enc_done = false;
    if (HAS_SEG_CHAIN && a.seg_chain) enc_done = NJ_CAT(njode_seg_chain_backward_, NJ_ID)(a, DROP, st) == hipSuccess && a.dw_enc_fused;
    else launch_ode_bwd_mfma<CC, DROP>(a, split, st);
    if (!enc_done) {
      ProfScope ps("k_encode_rows_bwd_mfma", st);
      k_encode_rows_bwd_mfma<CC, DROP><<<a.n_waves_rows / 4, 256, 0, st>>>(a);
    }
  }
}
template <class CC, bool DROP, bool TAIL>
static void launch_mfma_fwd(const KArgs& a, bool split, hipStream_t st) {
  if constexpr (HAS_MFMA) {
    const int n_tiles = cdiv(TAIL ? a.B : a.n_obs, 16);
    if constexpr (HAS_SPLIT) {
      if (split) {
        if constexpr (TAIL) {
          // (the four-wave form for every plan.  NJODE_TAILS=single: one wave per tile on the
          // scaled fragments for large plans, k_ode_fwd_tails -- measured beside the items' forward
          // on the autograd route and slower, 1.121 against 1.075 ms per step: 4 096 one-wave blocks
          // queue behind the forward's, the 1 024 four-wave blocks finish their tiles sooner)
          static const bool tails_single = getenv("NJODE_TAILS") && strcmp(getenv("NJODE_TAILS"), "single") == 0;
          if (n_tiles <= 768 || !tails_single)
            k_ode_fwd_split<CC, DROP, true><<<n_tiles < 1024 ? n_tiles : 1024, 256, 0, st>>>(a);
          else
            k_ode_fwd_tails<CC, DROP><<<n_tiles < 4096 ? n_tiles : 4096, 64, 0, st>>>(a);
        }
        else if (a.plan_job) {
          // the next batch's plan rides in front of this launch's own blocks (njode_plan.h)
          const PlanJob job = *(const PlanJob*)a.plan_job;
          // (never with NJODE_ENC_FUSED: that variant is two registers over the three-waves-per-SIMD
          // limit once it carries the plan; njode_api.hip launches the plan in front of such a call)
          k_ode_fwd_mixed_plan<CC, DROP, false><<<a.n_blocks_fwd + job.P, 256, 0, st>>>(a, job);
        }
        else if (a.enc_fused) k_ode_fwd_mixed<CC, DROP, true><<<a.n_blocks_fwd, 256, 0, st>>>(a);
        else k_ode_fwd_mixed<CC, DROP, false><<<a.n_blocks_fwd, 256, 0, st>>>(a);
        return;
      }
    }
    k_ode_fwd_mfma<CC, DROP, TAIL><<<n_tiles < 4096 ? n_tiles : 4096, 64, 0, st>>>(a);
  }
}

hipError_t NJ_CAT(njode_lock_forward_, NJ_ID)(const KArgs& a, bool drop, bool path, bool loss, int ode,
                                             hipStream_t st);
hipError_t NJ_CAT(njode_lock_backward_, NJ_ID)(const KArgs& a, bool drop, int ode, hipStream_t st);

#if NJ_PART == 0
template <bool DROP, bool TAIL, int ODE> static void launch_ode_fwd(const KArgs& a, hipStream_t st) {
  const int n_items = TAIL ? a.B : a.n_obs;
  if constexpr (ODE == ODE_MFMA) {
    launch_mfma_fwd<C, DROP, TAIL>(a, a.ode_split != 0, st);
  } else {
    constexpr bool WLDS = ODE == ODE_VALU_LDS;
    constexpr int NT = WLDS ? 256 : 64;
    k_ode_fwd_items<C, DROP, TAIL, WLDS><<<cdiv(n_items, NT), NT, 0, st>>>(a);
  }
}
template <bool DROP, int ODE>
static hipError_t seg_forward_t(const KArgs& a, bool tails, hipStream_t st) {
  if constexpr (C::MASKED || C::RNN) {
    return hipErrorNotSupported;
  } else {
    // pack + encoder rows: on the call's helper stream when there is one (they only need
    // t_of_row; the plan kernels already sit on `st`), else in line
    const SideInfo* side = (const SideInfo*)a.plan_ready;
    hipStream_t s2 = side ? side->st : st;
    if (side) (void)hipStreamWaitEvent(s2, side->e0, 0);
    // (a copy of the arguments: whether the keep bits are drawn ahead is decided here)
    KArgs ab = a;
    if constexpr (ODE == ODE_MFMA) {
      static const bool bits_off = getenv("NJODE_DROP_BITS_AHEAD") && atoi(getenv("NJODE_DROP_BITS_AHEAD")) == 0;
      // (plans whose every tile runs four waves wide, i.e. small batches: there the forward IS the
      // chain of its longest tile; in the mixed kernel of a large plan the four-wave blocks are
      // ~10 % of the work and the extra blocks of this launch cost more than they save:
      // 20 000 paths, k_pack_all 7.5 -> 12.3 us for ~1.5 us off k_ode_fwd_mixed)
      // ... and the wave-per-item forward's lane masks (they need nothing of the plan: any stream)
      const bool bits = a.seg_chain ? (DROP && a.dbits != nullptr)
                                    : (DROP && HAS_SPLIT && a.ode_split && a.dbits && !side && !bits_off &&
                                       a.n_split_fwd == a.n_blocks_fwd);
      ab.dbits_ready = bits ? 1 : 0;
      if (ab.plan_sync_zero && s2 != st) {   // (the pack launch is not on the hosting launch's stream)
        (void)hipMemsetAsync(ab.plan_sync_zero, 0, 8 * sizeof(unsigned), st);
        ab.plan_sync_zero = nullptr;
      }
      ProfScope ps("k_pack_all", s2);
      launch_pack_frags<C>(ab, s2, bits);
    }
    const bool enc_fused = ODE == ODE_MFMA && HAS_SPLIT && a.enc_fused != 0;
    if (!enc_fused) {
      ProfScope ps(ODE == ODE_MFMA ? "k_encode_rows_mfma" : "k_encode_rows", s2);
      if constexpr (ODE == ODE_MFMA) launch_mfma_enc<C, DROP>(a, s2);
      else k_encode_rows<C, DROP><<<cdiv(a.n_obs + a.B, 64), 64, 0, s2>>>(a);
    }
    if (side) {
      (void)hipEventRecord(side->e1, s2);
      (void)hipStreamWaitEvent(st, side->e1, 0);
    }
    if constexpr (ODE == ODE_MFMA && HAS_SPLIT) {
      if (enc_fused) {
        // NJODE_ENC_FUSED: what the ODE forward's one-wave role does not evaluate itself (needs the
        // plan -- the item order and the split point -- so it runs on `st`, behind it)
        ProfScope ps("k_encode_rows_items", st);
        const int n_path_tiles = cdiv(a.B, 16);
        k_encode_rows_items<C, DROP><<<n_path_tiles + a.n_split_fwd * 4, 64, 0, st>>>(a, n_path_tiles);
      }
    }
    // the tails (hT: every path from its last observation to the end) need the encoder's outputs
    // and nothing else of this call: with helper streams they start TOGETHER with the items' ODE
    // forward, on a stream of their own, and share the chip with it
    const bool chain = HAS_SEG_CHAIN && ODE == ODE_MFMA && a.seg_chain != 0;   // (tails ride in its launch)
    const bool tails_side = tails && side != nullptr && !chain;
    if (tails_side) {
      // the tails' stream waits for the encoder rows and for the plan's tail order: the latter is
      // on that stream itself (njode_api.hip, build_plan) or, failing that, on `st` -- then e0,
      // which the helper stream consumed above, is recorded again here to stand for "everything
      // `st` has enqueued by now"
      (void)hipStreamWaitEvent(side->st2, side->e1, 0);
      if (!side->tails_sorted_on_st2) {
        (void)hipEventRecord(side->e0, st);
        (void)hipStreamWaitEvent(side->st2, side->e0, 0);
      }
    }
    if (chain) {
      ProfScope ps("k_seg_fwd_chain", st);
      (void)NJ_CAT(njode_seg_chain_forward_, NJ_ID)(ab, DROP, tails, st);
    } else {   // (the names are the launched kernels', as rocprofv3 lists them)
      ProfScope ps(ODE == ODE_MFMA ? (HAS_SPLIT && a.ode_split ? "k_ode_fwd_mixed" : "k_ode_fwd_mfma")
                                   : "k_ode_fwd_items", st);
      launch_ode_fwd<DROP, false, ODE>(ab, st);
    }
    if (chain) {
    } else if (tails_side) {
      // (queued behind the forward's launch only so that the items' kernel is dispatched first)
      ProfScope ps(ODE == ODE_MFMA ? (HAS_SPLIT && a.ode_split ? "k_ode_fwd_split.tails" : "k_ode_fwd_mfma.tails")
                                   : "k_ode_fwd_items.tails", side->st2);
      launch_ode_fwd<DROP, true, ODE>(a, side->st2);
      (void)hipEventRecord(side->e2, side->st2);
    } else if (tails) {
      ProfScope ps(ODE == ODE_MFMA ? (HAS_SPLIT && a.ode_split ? "k_ode_fwd_split.tails" : "k_ode_fwd_mfma.tails")
                                   : "k_ode_fwd_items.tails", st);
      launch_ode_fwd<DROP, true, ODE>(a, st);
    }
    if (ODE == ODE_MFMA && a.defer_loss == 2) {
      // NJODE_C_ROWS_IN_FWD: the backward's row pass here (loss terms, readout gradients, adjoints
      // at the segment ends) instead of the forward-only pass
      launch_jump_rows_bwd<C, DROP>(a, st);
    } else if (!(ODE == ODE_MFMA && a.defer_loss)) {
      ProfScope ps(ODE == ODE_MFMA ? "k_jump_rows_mfma" : "k_jump_rows", st);
      if constexpr (ODE == ODE_MFMA) launch_mfma_jump<C, DROP>(a, st);
      else k_jump_rows<C, DROP><<<cdiv(a.n_obs, 64), 64, 0, st>>>(a);
    }
    if (tails_side) (void)hipStreamWaitEvent(st, side->e2, 0);
    return hipGetLastError();
  }
}
hipError_t NJ_CAT(njode_seg_forward_, NJ_ID)(const KArgs& a, bool drop, bool tails, int ode,
                                            hipStream_t st) {
  if (ode == ODE_MFMA && !HAS_MFMA) ode = ODE_VALU;
  switch (ode * 2 + (drop ? 1 : 0)) {
    case ODE_MFMA * 2 + 0: return seg_forward_t<false, ODE_MFMA>(a, tails, st);
    case ODE_MFMA * 2 + 1: return seg_forward_t<true, ODE_MFMA>(a, tails, st);
    case ODE_VALU_LDS * 2 + 0: return seg_forward_t<false, ODE_VALU_LDS>(a, tails, st);
    case ODE_VALU_LDS * 2 + 1: return seg_forward_t<true, ODE_VALU_LDS>(a, tails, st);
    case ODE_VALU * 2 + 1: return seg_forward_t<true, ODE_VALU>(a, tails, st);
    default: return seg_forward_t<false, ODE_VALU>(a, tails, st);
  }
}

const CfgOps* NJ_CAT(njode_cfg_ops_, NJ_ID)() {
  static const CfgOps ops = {
      {NJ_D, NJ_H, NJ_DO, NJ_NH, (NJ_NH > 0 ? NJ_W : 0), NJ_ACT,
       (NJ_MASKED ? NJODE_F_MASKED : 0) | (NJ_CURT ? NJODE_F_INPUT_CURRENT_T : 0) |
           (NJ_RES ? NJODE_F_RESIDUAL : 0) | (NJ_RNN ? NJODE_F_USE_RNN : 0)},
      C::P,
      C::ODE_IN,
      C::ENC_IN,
      C::OFF_ENC,
      C::OFF_DEC,
      NJ_CAT(njode_seg_forward_, NJ_ID),
      NJ_CAT(njode_seg_backward_, NJ_ID),
      NJ_CAT(njode_lock_forward_, NJ_ID),
      NJ_CAT(njode_lock_backward_, NJ_ID),
      MF_FLOATS,
      FS::ode,
      FS::ode + FS::enc,
      FRAG2_OFF,
      ACT_FLOATS,
      HAS_Q4 ? Q4_ACT_FLOATS : 0,
      HAS_MFMA_SWEEP ? 1 : 0,
      HAS_CHAIN ? 1 : 0,
      HAS_SEG_CHAIN ? 1 : 0,
      HAS_SPLIT ? 1 : 0,
      HAS_MFMA ? 1 : 0};
  return &ops;
}
#endif

#if NJ_PART == 1
template <bool DROP, int ODE> static hipError_t seg_backward_t(const KArgs& a, hipStream_t st) {
  if constexpr (C::MASKED || C::RNN) {
    return hipErrorNotSupported;
  } else if constexpr (ODE == ODE_MFMA) {
    launch_mfma_rows_bwd<C, DROP>(a, a.ode_split != 0, st);
    return hipGetLastError();
  } else {
    {
      ProfScope ps("k_jump_rows_bwd", st);
      k_jump_rows_bwd<C, DROP><<<a.n_waves, 64, 0, st>>>(a);
    }
    {
      ProfScope ps("k_ode_bwd_items", st);
      if constexpr (ODE == ODE_VALU_LDS) k_ode_bwd_items<C, DROP, true><<<a.n_waves / 4, 256, 0, st>>>(a);
      else k_ode_bwd_items<C, DROP, false><<<a.n_waves, 64, 0, st>>>(a);
    }
    {
      ProfScope ps("k_encode_rows_bwd", st);
      k_encode_rows_bwd<C, DROP><<<a.n_waves, 64, 0, st>>>(a);
    }
    return hipGetLastError();
  }
}
hipError_t NJ_CAT(njode_seg_backward_, NJ_ID)(const KArgs& a, bool drop, int ode, hipStream_t st) {
  if (ode == ODE_MFMA && !HAS_MFMA) ode = ODE_VALU;
  switch (ode * 2 + (drop ? 1 : 0)) {
    case ODE_MFMA * 2 + 0: return seg_backward_t<false, ODE_MFMA>(a, st);
    case ODE_MFMA * 2 + 1: return seg_backward_t<true, ODE_MFMA>(a, st);
    case ODE_VALU_LDS * 2 + 0: return seg_backward_t<false, ODE_VALU_LDS>(a, st);
    case ODE_VALU_LDS * 2 + 1: return seg_backward_t<true, ODE_VALU_LDS>(a, st);
    case ODE_VALU * 2 + 1: return seg_backward_t<true, ODE_VALU>(a, st);
    default: return seg_backward_t<false, ODE_VALU>(a, st);
  }
}
#endif

#if NJ_PART == 2
template <class CC> static void lock_pack_frags(const KArgs& a, hipStream_t st) {
  if constexpr (HAS_MFMA_LOCK) {
    using ES = typename EncS<CC>::type;
    using DS = typename DecS<CC>::type;
    k_pack_frags<CC><<<cdiv(MF<CC>::NALL * 64, 256), 256, 0, st>>>(a.P, a.frag);
    k_pack_net<typename CC::Enc, ES><<<cdiv(ES::NALL * 64, 256), 256, 0, st>>>(a.P + CC::OFF_ENC,
                                                                             a.frag_enc);
    k_pack_net<typename CC::Dec, DS><<<cdiv(DS::NALL * 64, 256), 256, 0, st>>>(a.P + CC::OFF_DEC,
                                                                             a.frag_dec);
  }
}
template <class CC, bool DROP> static void lock_launch_mfma(const KArgs& a, hipStream_t st) {
  if constexpr (HAS_CHAIN) {
    if (a.chain && !(a.want_path && DROP)) {   // (prediction calls: dropout-free ones)
      (void)NJ_CAT(njode_chain_forward_, NJ_ID)(a, DROP, st);
      return;
    }
  }
  if constexpr (HAS_Q4) {
    if (!a.want_path && lock4_on() && (!a.save_traj || a.lact)) {
      const int n_tiles = cdiv(a.B, a.q4_pt);
      KArgs ab = a;
      static const bool bits_off = getenv("NJODE_DROP_BITS_AHEAD") && atoi(getenv("NJODE_DROP_BITS_AHEAD")) == 0;
      if (DROP && a.dbits && a.dbits_row && !bits_off) {
        // the keep bits of every evaluation of the forward, drawn ahead in parallel over the chip
        const long long items = (long long)a.K * n_tiles;
        const int nb = (int)(items / 4 + 1 < 2048 ? items / 4 + 1 : 2048);
        k_q4_bits<CC><<<nb, 256, 0, st>>>(a, n_tiles);
        ab.dbits_ready = 1;
      }
      k_paths_fwd_q4<CC, DROP><<<n_tiles, 256, 0, st>>>(ab);
      return;
    }
  }
  if constexpr (HAS_MFMA_LOCK) k_paths_fwd_mfma<CC, DROP><<<cdiv(a.B, 32), 128, 0, st>>>(a);
}
template <bool DROP> static hipError_t lock_t(KArgs a, bool path, bool loss, int ode, hipStream_t st) {
  a.want_path = path ? 1 : 0;
  a.want_loss = loss ? 1 : 0;
  if (ode == ODE_MFMA && HAS_MFMA_LOCK) {
    // (the wave-per-path kernels read the flat parameter vector themselves)
    const bool chain = HAS_CHAIN && a.chain && !(a.want_path && DROP);
    if (!chain) lock_pack_frags<C>(a, st);
    ProfScope ps(chain ? "k_paths_fwd_chain" : "k_paths_fwd_mfma", st);
    lock_launch_mfma<C, DROP>(a, st);
  } else {
    ProfScope ps("k_paths_fwd", st);
    k_paths_fwd<C, DROP><<<cdiv(a.B, 64), 64, 0, st>>>(a);
  }
  return hipGetLastError();
}
hipError_t NJ_CAT(njode_lock_forward_, NJ_ID)(const KArgs& a, bool drop, bool path, bool loss, int ode,
                                             hipStream_t st) {
  return drop ? lock_t<true>(a, path, loss, ode, st) : lock_t<false>(a, path, loss, ode, st);
}
#endif

#if NJ_PART == 3
template <class CC, bool DROP> static void lock_bwd_mfma(const KArgs& a, hipStream_t st) {
  if constexpr (HAS_MFMA_SWEEP) {
    using ES = typename EncS<CC>::type;
    using DS = typename DecS<CC>::type;
    k_pack_frags<CC><<<cdiv(MF<CC>::NALL * 64, 256), 256, 0, st>>>(a.P, a.frag);
    k_pack_net<typename CC::Enc, ES><<<cdiv(ES::NALL * 64, 256), 256, 0, st>>>(a.P + CC::OFF_ENC,
                                                                             a.frag_enc);
    k_pack_net<typename CC::Dec, DS><<<cdiv(DS::NALL * 64, 256), 256, 0, st>>>(a.P + CC::OFF_DEC,
                                                                             a.frag_dec);
    {
      const bool chain = HAS_CHAIN && a.chain;
      ProfScope ps(chain ? "k_paths_bwd_adj_chain" : "k_paths_bwd_adj_mfma", st);
      bool q4 = false;
      if constexpr (HAS_CHAIN) {
        if (chain) {
          (void)NJ_CAT(njode_chain_sweep_, NJ_ID)(a, DROP, st);
          q4 = true;
        }
      }
      if constexpr (HAS_Q4) {
        if (!q4 && lock4_on() && a.lact) {   // (lact: the saving forward was k_paths_fwd_q4)
          k_paths_bwd_adj_q4<CC, DROP><<<cdiv(a.B, a.q4_pt), 256, 0, st>>>(a);
          q4 = true;
        }
      }
      if (!q4) k_paths_bwd_adj_mfma<CC, DROP><<<cdiv(a.B, 16), 64, 0, st>>>(a);
    }
    if (!NJ_CAT(njode_chain_dw_, NJ_ID)(a, st)) {
      ProfScope ps("k_ode_dw_pairs_mfma", st);
      k_ode_dw_pairs_mfma<CC, DROP><<<a.n_waves_rows / 4, 256, 0, st>>>(a);
    }
    {
      ProfScope ps("k_dec_dw_rows_mfma", st);
      k_dec_dw_rows_mfma<CC, DROP><<<a.n_waves_rows / 4, 256, 0, st>>>(a);
    }
    {
      ProfScope ps("k_enc_dw_rows_mfma", st);
      k_enc_dw_rows_mfma<CC, DROP><<<a.n_waves_rows / 4, 256, 0, st>>>(a);
    }
  }
}
// The one-lane-per-path sweep holds a path's whole state in registers: fine for the small
// unmasked shapes, but the 41-dimensional masked ones spill thousands of registers, so
// masked shapes are only differentiated on the matrix cores.
template <class CC, bool DROP> static hipError_t lock_bwd_valu(const KArgs& a, hipStream_t st) {
  if constexpr (CC::MASKED) {
    return hipErrorNotSupported;
  } else {
    {
      ProfScope ps("k_paths_bwd_adj", st);
      k_paths_bwd_adj<CC, DROP><<<cdiv(a.B, 64), 64, 0, st>>>(a);
    }
    {
      ProfScope ps("k_ode_dw_pairs", st);
      k_ode_dw_pairs<CC, DROP><<<a.n_waves, 64, 0, st>>>(a);
    }
    {
      ProfScope ps("k_dec_dw_rows", st);
      k_dec_dw_rows<CC, DROP><<<a.n_waves, 64, 0, st>>>(a);
    }
    {
      ProfScope ps("k_enc_dw_rows", st);
      k_enc_dw_rows<CC, DROP><<<a.n_waves, 64, 0, st>>>(a);
    }
    if constexpr (CC::RNN) {
      ProfScope ps("k_gru_dw_rows", st);
      k_gru_dw_rows<CC><<<a.n_waves, 64, 0, st>>>(a);
    }
    return hipGetLastError();
  }
}
template <bool DROP> static hipError_t lock_bwd_t(const KArgs& a, int ode, hipStream_t st) {
  if (ode == ODE_MFMA && HAS_MFMA_SWEEP) {
    lock_bwd_mfma<C, DROP>(a, st);
    return hipGetLastError();
  }
  return lock_bwd_valu<C, DROP>(a, st);
}
hipError_t NJ_CAT(njode_lock_backward_, NJ_ID)(const KArgs& a, bool drop, int ode, hipStream_t st) {
  return drop ? lock_bwd_t<true>(a, ode, st) : lock_bwd_t<false>(a, ode, st);
}
#endif

#if NJ_PART >= 4
// waves (= paths) per block: as few as still give every path a SIMD of its own
static inline int chain_waves_per_block(int B) {
  static const int env = getenv("NJODE_CHAIN_WPB") ? atoi(getenv("NJODE_CHAIN_WPB")) : 0;
  if (env >= 1 && env <= CHAIN_MAX_WAVES) return env;
  int w = 1;
  while (w < CHAIN_MAX_WAVES && cdiv(B, w) > 256) w *= 2;
  return w;
}
#endif
#if NJ_PART == 4
hipError_t NJ_CAT(njode_chain_forward_, NJ_ID)(const KArgs& a, bool drop, hipStream_t st) {
  if constexpr (HAS_CHAIN) {
    const int wpb = chain_waves_per_block(a.B);
    if (drop) {
      const long long items = (long long)a.K * a.B + (long long)a.n_obs * 3 + a.B;
      const int nb = (int)(items / 256 + 1 < 4096 ? items / 256 + 1 : 4096);
      k_chain_bits<C><<<nb, 256, 0, st>>>(a);
      k_paths_fwd_chain<C, true><<<cdiv(a.B, wpb), 64 * wpb, 0, st>>>(a);
    } else {
      k_paths_fwd_chain<C, false><<<cdiv(a.B, wpb), 64 * wpb, 0, st>>>(a);
    }
    return hipGetLastError();
  } else {
    return hipErrorNotSupported;
  }
}
#endif

#if NJ_PART == 4
hipError_t NJ_CAT(njode_seg_chain_forward_, NJ_ID)(const KArgs& a, bool drop, bool tails, hipStream_t st) {
  if constexpr (HAS_SEG_CHAIN) {
    const int nb = cdiv(a.n_obs + a.B, 4);
    if (drop && !a.dbits_ready) {
      const long long items = (long long)a.K * a.B;
      k_seg_chain_bits<C><<<(int)(items / 256 + 1 < 2048 ? items / 256 + 1 : 2048), 256, 0, st>>>(a);
    }
    if (a.plan_job) {
      // the next batch's plan rides in front of this launch's own blocks (njode_plan.h)
      const PlanJob job = *(const PlanJob*)a.plan_job;
      if (drop) k_seg_fwd_chain_plan<C, true><<<nb + job.P, 256, 0, st>>>(a, tails ? 1 : 0, job);
      else k_seg_fwd_chain_plan<C, false><<<nb + job.P, 256, 0, st>>>(a, tails ? 1 : 0, job);
    } else {
      if (drop) k_seg_fwd_chain<C, true><<<nb, 256, 0, st>>>(a, tails ? 1 : 0);
      else k_seg_fwd_chain<C, false><<<nb, 256, 0, st>>>(a, tails ? 1 : 0);
    }
    return hipGetLastError();
  } else {
    return hipErrorNotSupported;
  }
}
#endif

#if NJ_PART == 5
hipError_t NJ_CAT(njode_seg_chain_backward_, NJ_ID)(const KArgs& a, bool drop, hipStream_t st) {
  if constexpr (HAS_SEG_CHAIN) {
    {
      ProfScope ps("k_seg_bwd_chain", st);
      if (drop) k_seg_bwd_chain<C, true><<<cdiv(a.n_obs, 4), 256, 0, st>>>(a);
      else k_seg_bwd_chain<C, false><<<cdiv(a.n_obs, 4), 256, 0, st>>>(a);
    }
    // d loss / d ODE parameters: from the sweep's records (with the encoder's pass in the same launch),
    // else the lockstep plan's pair kernel on the stored adjoints
    if (a.dw_enc_fused) {
      if (!NJ_CAT(njode_chain_dw_enc_, NJ_ID)(a, drop, st)) return hipErrorLaunchFailure;
    } else if (!NJ_CAT(njode_chain_dw_, NJ_ID)(a, st)) {
      ProfScope ps("k_ode_dw_pairs_mfma", st);
      if (drop) k_ode_dw_pairs_mfma<C, true><<<a.n_waves_rows / 4, 256, 0, st>>>(a);
      else k_ode_dw_pairs_mfma<C, false><<<a.n_waves_rows / 4, 256, 0, st>>>(a);
    }
    return hipGetLastError();
  } else {
    return hipErrorNotSupported;
  }
}
bool NJ_CAT(njode_chain_dw_, NJ_ID)(const KArgs& a, hipStream_t st) {
  if constexpr ((HAS_CHAIN || HAS_SEG_CHAIN) && C::W < 64) {
    if (!(a.chain || a.seg_chain) || !a.cdelta || !a.cseg || a.dw_pair_blocks <= 0) return false;
    ProfScope ps("k_ode_dw_stored", st);
    k_ode_dw_stored<C><<<a.dw_pair_blocks + a.dw_seg_blocks, 256, 0, st>>>(a, a.dw_pair_blocks);
    return true;
  } else {
    return false;
  }
}
bool NJ_CAT(njode_chain_dw_enc_, NJ_ID)(const KArgs& a, bool drop, hipStream_t st) {
  if constexpr (HAS_SEG_CHAIN && C::W < 64) {
    if (!a.seg_chain || !a.cdelta || !a.cseg || a.dw_pair_blocks <= 0) return false;
    ProfScope ps("k_ode_dw_stored_enc", st);
    const int nb = a.dw_pair_blocks + a.dw_seg_blocks + a.n_waves_rows / 4;
    if (drop) k_ode_dw_stored_enc<C, true><<<nb, 256, 0, st>>>(a, a.dw_pair_blocks, a.dw_seg_blocks);
    else k_ode_dw_stored_enc<C, false><<<nb, 256, 0, st>>>(a, a.dw_pair_blocks, a.dw_seg_blocks);
    return true;
  } else {
    return false;
  }
}
hipError_t NJ_CAT(njode_chain_sweep_, NJ_ID)(const KArgs& a, bool drop, hipStream_t st) {
  if constexpr (HAS_CHAIN) {
    const int wpb = chain_waves_per_block(a.B);
    if (drop) k_paths_bwd_adj_chain<C, true><<<cdiv(a.B, wpb), 64 * wpb, 0, st>>>(a);
    else k_paths_bwd_adj_chain<C, false><<<cdiv(a.B, wpb), 64 * wpb, 0, st>>>(a);
    return hipGetLastError();
  } else {
    return hipErrorNotSupported;
  }
}
#endif

}  // namespace njode

#if defined(NJ_BWD_STAMPS) && NJ_PART == 1
// diagnostic build only (tools/ubench/bwd_stamps.sh): the per-wave stamps of k_ode_bwd_mixed
extern "C" int njode_debug_bwd_stamps(unsigned long long* dst, unsigned long long n_words) {
  const size_t cap = sizeof(njode::g_bwd_stamps) / 8;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(njode::g_bwd_stamps),
                                  (n_words < cap ? n_words : cap) * 8);
}
#endif
