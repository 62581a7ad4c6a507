// njode_host.h -- glue between the per-configuration translation units
// (njode_cfg.hip, one per compiled model shape) and the C-ABI unit (njode_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/njode_hip.h"
#include "njode_lockstep_bwd.h"
#include "njode_mfma_lockstep.h"
#include "njode_mfma_split.h"
#include "njode_mfma_lock4.h"

namespace njode {

constexpr int MAX_WAVES = 2048;
// ODE-evolve implementations: matrix cores (default where compiled), VALU with weights
// through the scalar cache, VALU with LDS-staged weights
constexpr int ODE_MFMA = 0, ODE_VALU = 1, ODE_VALU_LDS = 2;
// (NJODE_ODE=mfma1 keeps ODE_MFMA but with one wave per tile, njode_mfma.h, where the default
// picks the mixed kernels of njode_mfma_split.h: A/B baseline)

struct CfgOps {
  NjodeDims dims;
  int P;          // flat parameter count
  int ode_in, enc_in;
  int off_enc, off_dec;  // start of the encoder / readout slices in the flat vector
  // segment plan
  // tails: also evolve every path from its last observation to the end of the schedule
  // (hT); ode: implementation of the ODE-evolve kernels (ODE_*)
  hipError_t (*seg_forward)(const KArgs&, bool drop, bool tails, int ode, hipStream_t);
  hipError_t (*seg_backward)(const KArgs&, bool drop, int ode, hipStream_t);
  // lockstep plan
  // ode: ODE_MFMA runs the matrix-core lockstep kernel where the shape has one
  hipError_t (*lock_forward)(const KArgs&, bool drop, bool path, bool loss, int ode, hipStream_t);
  // ode: the implementation the saving forward ran (decides the dropout keying)
  hipError_t (*lock_backward)(const KArgs&, bool drop, int ode, hipStream_t);
  int frag_floats;  // size of the fragment buffer (0: no MFMA kernels for this shape)
  int frag_enc_off, frag_dec_off;  // offsets of the encoder / readout fragments in it
  int frag2_off;                   // ... of the scaled ODE table (njode_ode2.h)
  int act_floats;                  // stored ODE activations per chain and Euler step (0: none)
  int lock_act_floats;             // lockstep plan (masked shapes): ... per tile of 16 paths and step
  int lock_sweep_mfma;  // the lockstep backward has a matrix-core adjoint sweep
  int lock_chain;       // ... and the lockstep plan has the wave-per-path kernels (njode_chain.h)
  int seg_chain;        // the segment plan has the wave-per-item ODE kernels (njode_chain_seg.h)
  int ode_split;        // ODE_MFMA runs the mixed ODE kernels (njode_mfma_split.h)
  int seg_mfma;         // the segment plan has matrix-core kernels for this shape
};

// Helper stream of one forward call (KArgs::plan_ready points at it, host side only): the
// fragment packing and the encoder rows run there, BESIDE the plan kernels on the caller's
// stream -- the plan is the longer chain, so it stays where no cross-stream hop delays it.
// e0: recorded on the caller's stream once t_of_row exists (and everything before the call is
// done); e1: recorded on the helper stream after the encoder rows.
// st2 / e2 (round 4): a second helper stream of NORMAL priority for the tail items (hT: every path
// from its last observation to the end of the schedule).  They only need the encoder's outputs, so
// they start together with the ODE forward of the items and share the chip with it, instead of
// queueing behind it on the critical path of a training step that never reads hT.
struct SideInfo {
  hipStream_t st;
  hipEvent_t e0, e1;
  hipStream_t st2;
  hipEvent_t e2;
  int tails_sorted_on_st2;   // the plan's tail order was enqueued on st2 (else on the caller's stream)
};

// Optional per-kernel timing (njode_profile_enable / njode_profile_read): HIP events
// recorded on the launch stream around each kernel.  Defined in njode_api.hip.
void prof_mark(const char* name, hipStream_t st, bool begin);
struct ProfScope {
  const char* name;
  hipStream_t st;
  ProfScope(const char* n, hipStream_t s) : name(n), st(s) { prof_mark(name, st, true); }
  ~ProfScope() { prof_mark(name, st, false); }
};

}  // namespace njode
