// njode_cfg.hip -- one model shape, compiled once per entry of the build table
// (njode_amd/build.py) and per NJ_PART (0 segment forward + registration,
// 1 segment backward, 2 lockstep forward) with
//   -DNJ_ID=.. -DNJ_D=.. -DNJ_H=.. -DNJ_DO=.. -DNJ_NH=.. -DNJ_W=.. -DNJ_ACT=..
//   -DNJ_MASKED=.. -DNJ_CURT=.. -DNJ_RES=.. -DNJ_PART=..
#include "njode_host.h"

#define NJ_CAT_(a, b) a##b
#define NJ_CAT(a, b) NJ_CAT_(a, b)

namespace njode {

using C = Cfg<NJ_D, NJ_H, NJ_DO, NJ_NH, NJ_W, NJ_ACT, (NJ_MASKED != 0), (NJ_CURT != 0),
              (NJ_RES != 0)>;

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

hipError_t NJ_CAT(njode_seg_forward_, NJ_ID)(const KArgs& a, bool drop, bool tails, bool wlds,
                                            hipStream_t st);
hipError_t NJ_CAT(njode_seg_backward_, NJ_ID)(const KArgs& a, bool drop, bool wlds, hipStream_t st);
hipError_t NJ_CAT(njode_lock_forward_, NJ_ID)(const KArgs& a, bool drop, bool path, bool loss,
                                             hipStream_t st);

#if NJ_PART == 0
template <bool DROP, bool WLDS>
static hipError_t seg_forward_t(const KArgs& a, bool tails, hipStream_t st) {
  if constexpr (C::MASKED) {
    return hipErrorNotSupported;
  } else {
    constexpr int NT = WLDS ? 256 : 64;
    {
      ProfScope ps("k_encode_rows", st);
      k_encode_rows<C, DROP><<<cdiv(a.n_obs + a.B, 64), 64, 0, st>>>(a);
    }
    {
      ProfScope ps("k_ode_fwd_items", st);
      k_ode_fwd_items<C, DROP, false, WLDS><<<cdiv(a.n_obs, NT), NT, 0, st>>>(a);
    }
    if (tails) {
      ProfScope ps("k_ode_fwd_tails", st);
      k_ode_fwd_items<C, DROP, true, WLDS><<<cdiv(a.B, NT), NT, 0, st>>>(a);
    }
    {
      ProfScope ps("k_jump_rows", st);
      k_jump_rows<C, DROP><<<cdiv(a.n_obs, 64), 64, 0, st>>>(a);
    }
    return hipGetLastError();
  }
}
hipError_t NJ_CAT(njode_seg_forward_, NJ_ID)(const KArgs& a, bool drop, bool tails, bool wlds,
                                            hipStream_t st) {
  if (drop) return wlds ? seg_forward_t<true, true>(a, tails, st)
                        : seg_forward_t<true, false>(a, tails, st);
  return wlds ? seg_forward_t<false, true>(a, tails, st) : seg_forward_t<false, false>(a, tails, st);
}

const CfgOps* NJ_CAT(njode_cfg_ops_, NJ_ID)() {
  static const CfgOps ops = {
      {NJ_D, NJ_H, NJ_DO, NJ_NH, (NJ_NH > 0 ? NJ_W : 0), NJ_ACT,
       (NJ_MASKED ? NJODE_F_MASKED : 0) | (NJ_CURT ? NJODE_F_INPUT_CURRENT_T : 0) |
           (NJ_RES ? NJODE_F_RESIDUAL : 0)},
      C::P,
      C::ODE_IN,
      C::ENC_IN,
      NJ_CAT(njode_seg_forward_, NJ_ID),
      NJ_CAT(njode_seg_backward_, NJ_ID),
      NJ_CAT(njode_lock_forward_, NJ_ID)};
  return &ops;
}
#endif

#if NJ_PART == 1
template <bool DROP, bool WLDS> static hipError_t seg_backward_t(const KArgs& a, hipStream_t st) {
  if constexpr (C::MASKED) {
    return hipErrorNotSupported;
  } else {
    {
      ProfScope ps("k_jump_rows_bwd", st);
      k_jump_rows_bwd<C, DROP><<<a.n_waves, 64, 0, st>>>(a);
    }
    {
      ProfScope ps("k_ode_bwd_items", st);
      if constexpr (WLDS) k_ode_bwd_items<C, DROP, true><<<a.n_waves / 4, 256, 0, st>>>(a);
      else k_ode_bwd_items<C, DROP, false><<<a.n_waves, 64, 0, st>>>(a);
    }
    {
      ProfScope ps("k_encode_rows_bwd", st);
      k_encode_rows_bwd<C, DROP><<<a.n_waves, 64, 0, st>>>(a);
    }
    return hipGetLastError();
  }
}
hipError_t NJ_CAT(njode_seg_backward_, NJ_ID)(const KArgs& a, bool drop, bool wlds, hipStream_t st) {
  if (drop) return wlds ? seg_backward_t<true, true>(a, st) : seg_backward_t<true, false>(a, st);
  return wlds ? seg_backward_t<false, true>(a, st) : seg_backward_t<false, false>(a, st);
}
#endif

#if NJ_PART == 2
template <bool DROP> static hipError_t lock_t(KArgs a, bool path, bool loss, hipStream_t st) {
  ProfScope ps("k_paths_fwd", st);
  a.want_path = path ? 1 : 0;
  a.want_loss = loss ? 1 : 0;
  k_paths_fwd<C, DROP><<<cdiv(a.B, 64), 64, 0, st>>>(a);
  return hipGetLastError();
}
hipError_t NJ_CAT(njode_lock_forward_, NJ_ID)(const KArgs& a, bool drop, bool path, bool loss,
                                             hipStream_t st) {
  return drop ? lock_t<true>(a, path, loss, st) : lock_t<false>(a, path, loss, st);
}
#endif

}  // namespace njode
