/*
 * njode_hip.h -- C ABI of libnjode_hip.so: the MI355X (gfx950) NJ-ODE hot path.
 *
 * The reference (HerreraKrachTeichmann/NJODE) has no native code; its hot path is
 * the Python method NJODE.forward (NJODE/models.py:379-518) plus autograd's
 * backward and torch.optim.Adam (NJODE/train.py:397-398, 510-523).  This library
 * is what a native replacement of that path exports.  Each entry point names the
 * reference interface it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / STL types.
 *   - every pointer except `NjodeDims*`, `NjodeSchedule` host arrays and
 *     `size_t* out` is a DEVICE pointer; the caller owns every data buffer
 *     (parameters, batch, outputs, workspace, plan).
 *   - state the library keeps (all of it; none holds caller data):
 *       per thread   the error string of njode_last_error(); ONE pending plan job
 *                    (njode_plan_f32 with NJODE_C_PLAN_DEFER: a description, launched
 *                    by the thread's next njode_forward_f32 / njode_plan_f32 /
 *                    njode_plan_flush call);
 *       per process  two helper HIP streams and four events, created on first use,
 *                    for calls that build their plan in line (pack + encoder rows and
 *                    the hT tails run beside the plan); the kernel-timing records of
 *                    njode_profile_enable; the switches of DESIGN.md section 4g, each
 *                    read once from the environment;
 *       per device   64 bytes + 16 KB of device memory, allocated on first use: one
 *                    word that a grid barrier of the one-launch plan sets when it gives
 *                    up waiting (njode_plan_barrier_failures) and the stage stamps of
 *                    NJODE_PLAN_STAMPS=1.  The barrier COUNTERS of a plan are eight
 *                    words of that plan's own buffer.
 *   - all work is enqueued on the caller's `hipStream_t` (and, for in-line plans,
 *     on the helper streams, ordered against it by events); nothing synchronises
 *     except the functions that say so.
 *   - return 0 on success, an NJODE_E_* code otherwise (message via
 *     njode_last_error()).  No exceptions cross the ABI.
 *   - float data is fp32; indices are int32.
 *
 * Parameter vector (`params`, `grad_params`): one flat fp32 vector holding the
 * three networks in the reference's state_dict order (models.py:343-352;
 * SURVEY.md section 8 a12):
 *     ode_f.f.{0,3,6}.{weight,bias}, encoder_map.ffnn.{0,3,6}.{weight,bias},
 *     readout_map.ffnn.{0,3,6}.{weight,bias}
 *     [, obs_c.gru_d.{weight_ih,weight_hh,bias_ih,bias_hh} if use_rnn]
 * each weight in nn.Linear layout [out][in] row-major, each bias [out].  With
 * `bias=False` the bias slots are present and must be zero.
 */
#ifndef NJODE_HIP_H
#define NJODE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* njodeStream_t; /* == hipStream_t */

/* ---- error codes ----------------------------------------------------------- */
#define NJODE_OK 0
#define NJODE_E_UNSUPPORTED 1 /* no gfx950 specialisation for these dims/flags  */
#define NJODE_E_BADARG 2      /* null pointer / negative size / bad flag combo  */
#define NJODE_E_WORKSPACE 3   /* workspace too small                            */
#define NJODE_E_HIP 4         /* a HIP call or launch failed                    */

/* ---- model description (NJODE.__init__, models.py:284-362) ----------------- */
#define NJODE_ACT_TANH 0
#define NJODE_ACT_RELU 1

#define NJODE_F_MASKED 0x1           /* options['masked']           models.py:339-341 */
#define NJODE_F_INPUT_CURRENT_T 0x2  /* options['input_current_t']  models.py:334-337 */
#define NJODE_F_RESIDUAL 0x4         /* options['residual_enc_dec'] models.py:329-332 */
#define NJODE_F_LOSS_EASY 0x8        /* which_loss == 'easy'        models.py:109-126 */
#define NJODE_F_USE_RNN 0x10         /* use_rnn: GRU jump           models.py:202-217 */

#define NJODE_MAX_HIDDEN 8 /* hidden layers per network the library accepts (round 4: 8, was 4;
                              the reference's get_ffnn takes any depth, its grids use 1 - 2) */

/* One network of get_ffnn (models.py:140-166): nn_desc = ((width, act), ...). */
typedef struct NjodeNet {
  int32_t n_hidden;                 /* len(nn_desc); 0 for nn_desc = None (one Linear)   */
  int32_t width[NJODE_MAX_HIDDEN];  /* width of hidden layer l                            */
  int32_t act[NJODE_MAX_HIDDEN];    /* NJODE_ACT_* behind hidden layer l                  */
} NjodeNet;

typedef struct NjodeDims {
  int32_t input_size;  /* d                                                      */
  int32_t hidden_size; /* H                                                      */
  int32_t output_size; /* d_out                                                  */
  int32_t n_hidden;    /* hidden layers of each of the three nets (len(nn_desc)) */
  int32_t width;       /* width of those hidden layers (ignored if n_hidden==0)  */
  int32_t act;         /* NJODE_ACT_*                                            */
  int32_t flags;       /* NJODE_F_*                                              */
  /* per_net = 0: ode_nn, enc_nn and readout_nn all are n_hidden layers of `width` units with
   * `act` (every configuration of the reference's own scripts).  per_net = 1: the three
   * descriptions follow in nets[] (ode_nn, enc_nn, readout_nn) and n_hidden / width / act are
   * ignored -- networks that differ from each other, layers of different widths or
   * activations, up to NJODE_MAX_HIDDEN hidden layers. */
  int32_t per_net;
  NjodeNet nets[3];
} NjodeDims;

/* ---- per-call options ------------------------------------------------------- */
#define NJODE_C_TRAIN 0x1       /* model.train(): dropout active                  */
#define NJODE_C_GET_LOSS 0x2    /* get_loss=True                                  */
#define NJODE_C_RETURN_PATH 0x4 /* return_path=True                               */
#define NJODE_C_SAVE_BWD 0x8    /* keep what njode_backward_f32 needs in workspace */
#define NJODE_C_LOSS_IN_BWD 0x10 /* fused step: the loss may be produced by the backward call
                                   (njode_backward_loss_f32, same `loss` pointer) instead of
                                   the forward call -- saves one pass over the observation rows */
#define NJODE_C_SCHED_KNOWN 0x20 /* the caller states whether the schedule has a tail (Euler steps
                                   after the last jump) in NJODE_C_SCHED_TAIL; the library then
                                   never reads the host schedule arrays to choose the plan, so
                                   njode_backward*_f32 may be called after they were reused     */
#define NJODE_C_PLAN_READY 0x80  /* NjodeBatch.plan holds the schedule copy and the execution plan,
                                   built ahead of time by njode_plan_f32 (below)              */
#define NJODE_C_NEED_HT 0x100    /* (njode_plan_f32) the forward will be asked for hT         */
#define NJODE_C_SCHED_TAIL 0x40  /* (with SCHED_KNOWN) k_jump[n_times-1] < n_steps             */
#define NJODE_C_ROWS_IN_FWD 0x400 /* (with SAVE_BWD, segment plan on the matrix cores) the forward runs the
                                   backward's pass over the observation rows itself -- both readouts of
                                   every row, the loss terms AND their adjoints -- instead of a
                                   forward-only pass; njode_backward_f32 of this call then skips it.
                                   For callers whose saving forward is always followed by its backward
                                   (an autograd bridge): the readouts are evaluated once, not twice.
                                   Same value to the forward and its backward.                        */
#define NJODE_C_PLAN_DEFER 0x800 /* (njode_plan_f32) do not launch anything now: the plan is built by the
                                   first blocks of the ODE-forward launch of the NEXT njode_forward_f32
                                   call this host thread makes on the same stream (one queue, no events,
                                   no dispatches of its own: see njode_plan_f32 below)               */
#define NJODE_C_GEN_LOCKSTEP 0x200 /* keep an unmasked loss call on the LOCKSTEP plan (both kernel
                                   families since round 5; round 4: the shape-generic one): A/B runs,
                                   tests, and the pass that differentiates through hT
                                   (NjodeBatch.grad_hT).  A CALL flag -- the same value must be
                                   given to njode_plan_f32, the forward and its backward, which lay
                                   the plan and the workspace out by it -- not an environment
                                   variable the library would re-read in each of the three       */

/*
 * Time grid of one forward pass: the float64 clock of NJODE.forward
 * (models.py:430-439, 497-505) evaluated on the host and rounded exactly as
 * ATen rounds it (python-float operands are cast to fp32).  HOST arrays; the
 * library copies them to the device workspace asynchronously, so they must stay
 * valid until the stream has passed the call (use pinned memory).
 */
typedef struct NjodeSchedule {
  int32_t n_steps;        /* K: Euler steps in this pass (incl. the until_T tail)    */
  int32_t n_times;        /* number of observation times (len(times))                */
  const float* step_dt;   /* [K]  fp32(delta_t_k)                                    */
  const float* step_t;    /* [K]  fp32(current_time before step k)                   */
  const int32_t* k_jump;  /* [n_times] Euler steps completed before jump i happens   */
  const float* time_f32;  /* [n_times] fp32(times[i])  (the value written to tau)    */
  const int32_t* time_ptr;/* [n_times+1] CSR offsets into X / obs_idx (models.py:449)*/
} NjodeSchedule;

/* One batch in the layout of data_utils.custom_collate_fn (data_utils.py:278-316). */
typedef struct NjodeBatch {
  int32_t batch_size;      /* B: paths held by this call (this rank's shard)         */
  int32_t n_obs;           /* rows of X / obs_idx                                     */
  const float* start_X;    /* [B, d]                                                  */
  const float* X;          /* [n_obs, d]  sorted by time, then path                   */
  const float* M;          /* [n_obs, d]  0/1 mask, or NULL (required iff MASKED)     */
  const int32_t* obs_idx;  /* [n_obs]     path index of each row, in [0, B); within one time   */
                           /*             slice a path has AT MOST ONE row (both collates of    */
                           /*             the reference guarantee it).  Not checked unless the  */
                           /*             environment has NJODE_VALIDATE=1 (then: one extra      */
                           /*             kernel + a sync per forward, NJODE_E_BADARG on a bad  */
                           /*             index, a duplicate, or n_obs_ot == 0 for an observed  */
                           /*             path)                                                */
  const int32_t* n_obs_ot; /* [B] observations per path, or NULL if !GET_LOSS         */
  float loss_batch_size;   /* the `batch_size` in compute_loss (models.py:106); for a */
                           /* data-parallel shard pass the GLOBAL batch size          */
  int64_t path_id_offset;  /* global id of path 0 (dropout streams are keyed by it)   */
  const void* plan;        /* NJODE_C_PLAN_READY: the buffer njode_plan_f32 filled for */
                           /* this batch, schedule and call_flags; else ignored (NULL) */
  const float* grad_hT;    /* njode_backward_f32 only, or NULL: [B, H] upstream gradient of  */
                           /* hT (the reference returns hT inside its autograd graph,        */
                           /* models.py:414-518).  Added to the adjoint of the final state   */
                           /* by the LOCKSTEP plan's sweep -- i.e. the call must run that    */
                           /* plan (masked / use_rnn models, schedules with a tail, or       */
                           /* NJODE_C_GEN_LOCKSTEP); a segment-plan backward returns         */
                           /* NJODE_E_UNSUPPORTED for a non-NULL value.  grad_params then    */
                           /* is grad_loss * (d loss / d params + d <grad_hT, hT> / d params) */
} NjodeBatch;

/* ---- queries ----------------------------------------------------------------- */

/* 1 if the library runs `dims`, else 0.  Two kernel families stand behind the same entry
 * points: shape-specialised kernels for the shapes of the build table (njode_build_info lists
 * them: the demo / PhysioNet / convergence-study shapes with widths < 64, the demo shape with
 * the GRU jump), and the shape-generic matrix-core kernels (njode_gen.h) for everything else:
 * any sizes, widths up to NJODE_GEN_MAX_WIDTH (a GRU cell: 4 x hidden_size), per-network
 * descriptions (per_net = 1), use_rnn on unmasked models. */
#define NJODE_GEN_MAX_WIDTH 1024
int njode_supported(const NjodeDims* dims);

/* Length P of the flat parameter vector for `dims` (0 if unsupported). */
size_t njode_param_count(const NjodeDims* dims);

/* Bytes of workspace njode_forward_f32 / njode_backward_f32 need for a batch of
 * this size (upper bound; n_steps/n_times as in NjodeSchedule).  With NJODE_C_SAVE_BWD the
 * workspace also holds what the backward reads back: the state before every Euler step
 * (H floats per path and step) and -- matrix-core shapes -- the ODE network's hidden
 * activations of every step (8 ceil((width + 1) / 4) floats per path and step, 416 B for
 * width 50: 845 MB for 20 000 paths x 100 steps; masked shapes 8 KB per 16 paths and step).
 * The same call_flags must be passed here, to the forward and to the backward of one step. */
int njode_workspace_bytes(const NjodeDims* dims, int32_t batch_size, int32_t n_obs,
                          int32_t n_times, int32_t n_steps, int32_t call_flags,
                          size_t* out);

/*
 * Plan ahead (optional).  The execution plan of a call -- device copy of the schedule, rows
 * linked per path, segments sorted by length, trajectory layout: ~0.12 ms of small
 * latency-bound kernels at 20 000 paths -- depends on the batch and the schedule only, not on
 * the parameters.  A training loop that has batch i+1 in hand while step i runs can build it on
 * another stream, beside step i's kernels: njode_plan_f32 writes it into a caller-owned buffer
 * of njode_plan_bytes bytes, and the forward AND the backward of that batch are then called
 * with NJODE_C_PLAN_READY and NjodeBatch.plan = that buffer (which they only read), skipping
 * the plan stage.  Same dims, batch arrays, schedule and call_flags (NJODE_C_PLAN_READY aside)
 * as those calls; NJODE_C_NEED_HT if the forward will be given hT != NULL.  The host schedule
 * arrays must stay valid until `stream` has executed the call (an asynchronous copy).
 *
 * NJODE_C_PLAN_DEFER (round 5): a second queue costs the step it runs beside two event hand-overs
 * and a dozen small dispatches (32 us of an 880 us step at 20 000 paths).  With this flag the
 * call only describes the job; the next njode_forward_f32 call of this host thread on the SAME
 * stream carries it as the first blocks of its ODE-forward launch (segment plan on the matrix
 * cores), or launches it as one kernel in front of itself when it cannot (another kernel family,
 * another stream, the very call that consumes this plan).  Call order: njode_plan_f32(batch i+1,
 * DEFER) ... njode_forward_f32(batch i) ... njode_forward_f32(batch i+1, PLAN_READY): stream order
 * does the rest.  The schedule arrays are then read by the device from pinned host memory (else
 * copied at once) and must stay valid until the stream has passed the hosting forward call.
 * njode_plan_flush launches a job that is still pending at once (0 = nothing was pending).  Plans
 * the single launch does not build (K >= 512 steps, NJODE_VALIDATE, lockstep plans) are built
 * immediately, as without the flag.
 */
int njode_plan_flush(void);
int njode_plan_bytes(const NjodeDims* dims, int32_t batch_size, int32_t n_obs,
                     int32_t n_times, int32_t n_steps, int32_t call_flags, size_t* out);
int njode_plan_f32(const NjodeDims* dims, const NjodeBatch* batch, const NjodeSchedule* sched,
                   int32_t call_flags, void* plan, size_t plan_bytes, njodeStream_t stream);

/* ---- the hot path ------------------------------------------------------------ */

/*
 * Replaces NJODE.forward (models.py:379-518): encoder -> Euler ODE-evolve between
 * observation times -> jump at observations -> readout -> paper loss
 * (models.py:71-126).
 *
 *   hT       [B, H]            hidden state at the end of the pass (every path evolved
 *                              to the last step of the schedule, as the reference
 *                              does); may be NULL on the segment plan, which then
 *                              skips the per-path tail evolve nobody reads
 *   loss     [1]               (written iff GET_LOSS) sum over this shard's rows
 *   path_h   [n_rows, B, H]    (iff RETURN_PATH) n_rows = 1 + n_steps + n_times
 *   path_y   [n_rows, B, d_out]
 *
 * Two execution plans, chosen by the library:
 *   - segment plan (unmasked, no RETURN_PATH, schedule ends at the last
 *     observation): every (path, inter-observation segment) is an independent
 *     work item; items are sorted by length, 16 of (almost) equal length form a
 *     tile, a tile runs on one wave (or, the longest ones, on the four waves of a
 *     block) of the matrix-core kernels.
 *   - lockstep plan (everything else): 16 paths per wave over the shared grid,
 *     jumps applied under a wave ballot.
 * `weight` is NJODE.weight (models.py:316), `dropout_p` the dropout rate,
 * `seed` the dropout stream seed (only read when TRAIN and dropout_p > 0).
 */
int njode_forward_f32(const NjodeDims* dims, const float* params,
                      const NjodeBatch* batch, const NjodeSchedule* sched,
                      int32_t call_flags, float weight, float dropout_p,
                      uint64_t seed, float* hT, float* loss, float* path_h,
                      float* path_y, void* workspace, size_t workspace_bytes,
                      njodeStream_t stream);

/*
 * Replaces loss.backward() (train.py:522) for the graph NJODE.forward built:
 * the exact discrete adjoint of the Euler/jump recursion.  Must follow a
 * njode_forward_f32 call with NJODE_C_SAVE_BWD | NJODE_C_GET_LOSS on the same
 * workspace, batch, schedule, params, weight, dropout_p and seed.
 *
 *   grad_loss    [1]   upstream gradient of the scalar loss (device)
 *   grad_params  [P]   OVERWRITTEN with d loss / d params * grad_loss
 *
 * Both plans are covered: the segment plan runs the reverse sweep per segment on the
 * matrix cores; the lockstep plan (masked models, schedules with a tail) runs an adjoint
 * sweep per path followed by parallel weight-gradient kernels.
 */
int njode_backward_f32(const NjodeDims* dims, const float* params,
                       const NjodeBatch* batch, const NjodeSchedule* sched,
                       int32_t call_flags, float weight, float dropout_p,
                       uint64_t seed, const float* grad_loss, float* grad_params,
                       void* workspace, size_t workspace_bytes,
                       njodeStream_t stream);

/*
 * Fused training step (model(...) followed by loss.backward(), train.py:510-522, with
 * grad_loss = 1): njode_forward_f32(..., flags | NJODE_C_LOSS_IN_BWD, ..., loss, ...) followed
 * by this call with the same flags and the same `loss` pointer.  Where the plan allows it the
 * forward skips its readout/loss pass over the observation rows and this call, which
 * evaluates the same readouts for their gradients anyway, writes the loss; otherwise the
 * forward has already written it and this call leaves it alone.
 */
int njode_backward_loss_f32(const NjodeDims* dims, const float* params,
                            const NjodeBatch* batch, const NjodeSchedule* sched,
                            int32_t call_flags, float weight, float dropout_p,
                            uint64_t seed, const float* grad_loss, float* grad_params,
                            float* loss, void* workspace, size_t workspace_bytes,
                            njodeStream_t stream);

/*
 * Replaces torch.optim.Adam(lr, betas, eps, weight_decay).step() on the flat
 * parameter vector (train.py:397-398, 523): L2 weight decay folded into the
 * gradient, bias-corrected moments, `step` = 1-based step count.
 * `grad_scale` multiplies the gradient first (e.g. 1 for summed DP shards).
 */
int njode_adam_step_f32(float* params, const float* grad, float* exp_avg,
                        float* exp_avg_sq, size_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int32_t step,
                        float grad_scale, njodeStream_t stream);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* njode_last_error(void);

/*
 * Measurement aid (not part of the replaced reference surface): when enabled,
 * every hot-path kernel launch is bracketed by hipEvents recorded on the launch
 * stream.  njode_profile_read synchronises the device and writes one line per
 * kernel, "<name> <launches> <total_ms>\n", then clears the records (process-global
 * state: see "state the library keeps" at the top).  on = 1: every kernel (twelve events per training step,
 * ~3.5 % of a 1.1 ms step); on = 2: the ODE backward kernel only (two events).
 */
int njode_profile_enable(int on);
int njode_profile_read(char* out, size_t cap);

/*
 * 1 when a grid barrier of a one-launch plan (njode_plan_f32, or the plan an njode_forward_f32 call
 * built in line) gave up waiting since the last call -- the plans built since then are invalid and
 * must be rebuilt; njode_last_error() says so -- 0 otherwise, -1 on a HIP error.  The barrier's spin is
 * bounded (seconds), so a lost block shows up here instead of as a hung device.  SYNCHRONISES the
 * device: for tests and NJODE_VALIDATE-style checks, not for the hot loop.
 */
int njode_plan_barrier_failures(void);

/* Build information: "gfx950;<list of compiled specialisations>". */
const char* njode_build_info(void);


#ifdef __cplusplus
}
#endif
#endif /* NJODE_HIP_H */
