// njode_gen.hip -- host side of the shape-generic kernel family (njode_gen.h): model
// description -> layer tables, workspace layout, plan (dense [time][path] -> row matrix, first
// jump of every Euler step), launches.  Entry points are the gen_* functions below; the C ABI
// (njode_api.hip) routes to them every model shape the build table has no specialisation for.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <rocprim/rocprim.hpp>

#include "../../include/njode_hip.h"
#include "njode_error.h"
#include "njode_gen.h"
#include "njode_gen_seg.h"
#include "njode_gen_host.h"

namespace njode {
void prof_mark(const char* name, hipStream_t st, bool begin);   // njode_api.hip
}

using namespace njode;
using namespace njode::gen;

namespace {

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  njode::set_error_v(code, fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess)                                                          \
      return fail(NJODE_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));     \
  } while (0)

struct Prof {
  const char* name;
  hipStream_t st;
  Prof(const char* n, hipStream_t s) : name(n), st(s) { njode::prof_mark(name, st, true); }
  ~Prof() { njode::prof_mark(name, st, false); }
};

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

constexpr int LDS_LIMIT = 160 * 1024;

struct Model {
  GArgs a;          // dims and layer tables filled in; pointers null
  int P;            // flat parameter count
  int frag_floats;
  int nw;           // waves per workgroup of the forward / sweep kernels
  int lds_bytes;
  int S;            // slab rows (K splits of the weight-gradient GEMMs)
  GPack pack;
};

// nets of a model in the order of the flat parameter vector (state_dict order)
bool build_net(GNet& N, int n_in, int n_out, const NjodeNet& d, int& p_off, int& f_off,
               GPack& pack, int& img_rows, int& max_mt, int& max_tb) {
  if (d.n_hidden < 0 || d.n_hidden > NJODE_MAX_HIDDEN) return false;
  N.nl = d.n_hidden + 1;
  N.n_in = n_in;
  N.n_out = n_out;
  int a_row = 0;
  for (int l = 0; l < N.nl; ++l) {
    GLayer& L = N.l[l];
    L.n_in = l == 0 ? n_in : d.width[l - 1];
    L.n_out = l == d.n_hidden ? n_out : d.width[l];
    if (L.n_in <= 0 || L.n_out <= 0 || L.n_in > NJODE_GEN_MAX_WIDTH + 64 || L.n_out > NJODE_GEN_MAX_WIDTH)
      return false;
    L.act = l < d.n_hidden ? d.act[l] : -1;
    if (l < d.n_hidden && L.act != NJODE_ACT_TANH && L.act != NJODE_ACT_RELU) return false;
    L.w_off = p_off;
    L.b_off = p_off + L.n_in * L.n_out;
    p_off = L.b_off + L.n_out;
    L.Qp = pad_to(cdiv(L.n_in + 1, 4), QU);
    L.MT = cdiv(L.n_out, 16);
    L.QTp = pad_to(cdiv(L.n_out, 4), QU);
    L.MTT = cdiv(L.n_in, 16);
    L.f_off = f_off;
    f_off += L.MT * L.Qp * 64;
    L.ft_off = f_off;
    f_off += L.MTT * L.QTp * 64;
    L.a_row = a_row;
    a_row += L.n_in;
    int need = 4 * L.Qp;
    if (4 * L.QTp > need) need = 4 * L.QTp;
    if (L.n_in + 1 > need) need = L.n_in + 1;
    if (L.n_out + 1 > need) need = L.n_out + 1;
    if (need > img_rows) img_rows = need;
    if (L.MT > max_mt) max_mt = L.MT;
    if (L.MTT > max_mt) max_mt = L.MTT;
    const int tb = cdiv(L.MT, DW_TM) * cdiv(cdiv(L.n_in + 1, 16), DW_TN);
    if (tb > max_tb) max_tb = tb;
  }
  int d_row = a_row;
  for (int l = 0; l < N.nl; ++l) {
    N.l[l].d_row = d_row;
    d_row += N.l[l].n_out;
  }
  N.rec_rows = d_row;
  for (int l = 0; l < N.nl; ++l) pack.l[pack.n_layers++] = N.l[l];
  return true;
}

bool build_model(const NjodeDims* d, Model& m, const char** why) {
  static const char* none = "";
  *why = none;
  memset(&m, 0, sizeof(m));
  if (!d) { *why = "null dims"; return false; }
  const int D = d->input_size, H = d->hidden_size, DO = d->output_size;
  if (D <= 0 || H <= 0 || DO <= 0 || D > 512 || H > 1024 || DO > 512) { *why = "sizes out of range"; return false; }
  const bool masked = d->flags & NJODE_F_MASKED, curt = d->flags & NJODE_F_INPUT_CURRENT_T,
             res = d->flags & NJODE_F_RESIDUAL;
  // (round 5) output_size != input_size: the reference builds a readout to any output_size
  // (models.py:350-352); what compares X with the readout -- the loss, and the masked mode's
  // self-imputation / prediction feed-back (models.py:465-467, 483-484) -- needs equal sizes, so
  // such a model runs prediction calls only (get_loss = False; checked per call in prepare())
  if (D != DO && masked) { *why = "masked mode feeds the readout back as the input: input_size must equal output_size"; return false; }
  GArgs& a = m.a;
  a.D = D; a.H = H; a.DO = DO;
  a.masked = masked; a.curt = curt;
  a.loss_easy = (d->flags & NJODE_F_LOSS_EASY) ? 1 : 0;
  a.IN0 = D + H + (curt ? 3 : 2);
  a.enc_case = a.dec_case = 0;
  a.enc_mult = a.dec_mult = 1;
  if (res) {   // FFNN residual rules (models.py:240-259)
    if (D <= H) { if (H % D) { *why = "residual: output_size needs to be multiple of input_size"; return false; } a.enc_case = 1; a.enc_mult = H / D; }
    else { if (D % H) { *why = "residual: input_size needs to be multiple of output_size"; return false; } a.enc_case = 2; a.enc_mult = D / H; }
    if (H <= DO) { if (DO % H) { *why = "residual: output_size needs to be multiple of input_size"; return false; } a.dec_case = 1; a.dec_mult = DO / H; }
    else { if (H % DO) { *why = "residual: input_size needs to be multiple of output_size"; return false; } a.dec_case = 2; a.dec_mult = H / DO; }
  }
  NjodeNet nets[3];
  if (d->per_net) {
    for (int i = 0; i < 3; ++i) nets[i] = d->nets[i];
  } else {
    if (d->n_hidden < 0 || d->n_hidden > NJODE_MAX_HIDDEN) { *why = "n_hidden out of range"; return false; }
    for (int i = 0; i < 3; ++i) {
      nets[i].n_hidden = d->n_hidden;
      for (int l = 0; l < NJODE_MAX_HIDDEN; ++l) { nets[i].width[l] = d->width; nets[i].act[l] = d->act; }
    }
  }
  int p_off = 0, f_off = 0, img_rows = 16, max_mt = 1, max_tb = 1;
  m.pack.n_layers = 0;
  if (!build_net(a.ode, a.IN0, H, nets[0], p_off, f_off, m.pack, img_rows, max_mt, max_tb) ||
      !build_net(a.enc, masked ? 2 * D : D, H, nets[1], p_off, f_off, m.pack, img_rows, max_mt, max_tb) ||
      !build_net(a.dec, H, DO, nets[2], p_off, f_off, m.pack, img_rows, max_mt, max_tb)) {
    *why = "network description out of range (<= 8 hidden layers, widths <= 1024, tanh / relu)";
    return false;
  }
  a.rnn = (d->flags & NJODE_F_USE_RNN) ? 1 : 0;
  if (a.rnn) {
    // the GRU cell as ONE layer [x; h; 1] -> [r, z, W_in x + b_in, W_hn h + b_hn]  (njode_gen.h, gru_w)
    // (round 5: with masked data the reference runs the cell on the zero-filled X_obs -- no
    // self-imputation at the jump -- while the start state still goes through the masked encoder,
    // the loss is masked and last_X <- Y (models.py:411-414, 460-461, 483-484); the lockstep kernels
    // of this family are written per feature, so the combination needs no code of its own)
    if (4 * H > NJODE_GEN_MAX_WIDTH) { *why = "use_rnn: 4 x hidden_size exceeds the widest layer the kernels take"; return false; }
    GNet1& N = a.gru;
    N.nl = 1;
    N.n_in = D + H;
    N.n_out = 4 * H;
    GLayer& L = N.l[0];
    memset(&L, 0, sizeof(L));
    L.kind = 1;
    L.n_in = D + H;
    L.n_out = 4 * H;
    L.act = -1;
    L.w_off = p_off;                              // weight_ih, weight_hh, bias_ih, bias_hh
    L.b_off = p_off + 3 * H * D + 3 * H * H;
    p_off += 3 * H * D + 3 * H * H + 6 * H;
    L.Qp = pad_to(cdiv(L.n_in + 1, 4), QU);
    L.MT = cdiv(L.n_out, 16);
    L.QTp = pad_to(cdiv(L.n_out, 4), QU);
    L.MTT = cdiv(L.n_in, 16);
    L.f_off = f_off;
    f_off += L.MT * L.Qp * 64;
    L.ft_off = f_off;
    f_off += L.MTT * L.QTp * 64;
    L.a_row = 0;
    L.d_row = L.n_in;
    N.rec_rows = L.n_in + L.n_out;
    int need = 4 * L.Qp;
    if (4 * L.QTp > need) need = 4 * L.QTp;
    if (L.n_out + 1 > need) need = L.n_out + 1;
    if (need > img_rows) img_rows = need;
    if (L.MT > max_mt) max_mt = L.MT;
    const int tb = cdiv(L.MT, DW_TM) * cdiv(cdiv(L.n_in + 1, 16), DW_TN);
    if (tb > max_tb) max_tb = tb;
    m.pack.l[m.pack.n_layers++] = L;
  }
  m.P = p_off;
  m.frag_floats = f_off + RING * 256;   // (+ one chunk: the product loop reads whole chunks)
  m.pack.total = f_off;
  // the vectors the kernels stage in the images besides layer inputs: ODE input, readouts, states
  if (a.IN0 + 4 > img_rows) img_rows = a.IN0 + 4;
  if (2 * D + 4 > img_rows) img_rows = 2 * D + 4;
  if (H + 4 > img_rows) img_rows = H + 4;
  a.img_rows = pad_to(img_rows + 4, 4);
  m.lds_bytes = gen_lds_floats(a.img_rows, D, H, DO) * 4;
  if (m.lds_bytes > LDS_LIMIT) { *why = "layer images exceed the 160 KB LDS"; return false; }
  m.nw = max_mt < 4 ? 4 : (max_mt > 16 ? 16 : max_mt);
  if (const char* e = getenv("NJODE_GEN_NW")) {        // experiments: waves per workgroup
    const int v = atoi(e);
    if (v >= 1 && v <= 16) m.nw = v;
  }
  m.S = 2048 / max_tb;
  if (m.S < 8) m.S = 8;
  if (m.S > 256) m.S = 256;
  return true;
}

struct Layout {
  size_t total = 0;
  size_t sched, jlo, dense, bad, plan_end;
  size_t frag, loss_terms, slab, rec_ode, rec_enc, rec_dec, rec_gru = 0, ybuf, flags;
  // segment plan (unmasked loss calls): plan prefix ...
  size_t t_of_row, item_prev, item_next, item_kbeg, item_len, key, key_sorted, iota, order,
      first_row, last_row, tail_key, tail_key_sorted, iota_b, tail_order, tile_base, sort_tmp;
  size_t sort_tmp_bytes = 0;
  // ... and per-call arrays
  size_t h0row, h0start, h_end, lam_end, g_h0, lam_start;
  int T, PT = 16;   // path tiles and paths per tile
  int NT = 0;       // item tiles = row tiles
  bool seg = false;
  size_t take(size_t bytes) {
    size_t off = total;
    total += (bytes + 255) & ~(size_t)255;
    return off;
  }
};

inline int bits_for(unsigned v) {  // radix bits needed for keys in [0, v]
  int b = 1;
  while (b < 32 && (v >> b)) ++b;
  return b;
}
hipError_t sort_pairs(void* tmp, size_t& bytes, const unsigned* k_in, unsigned* k_out, const int* v_in,
                      int* v_out, int n, int end_bit, hipStream_t st) {
  return rocprim::radix_sort_pairs(tmp, bytes, k_in, k_out, v_in, v_out, (size_t)(n > 0 ? n : 1), 0u,
                                   (unsigned)end_bit, st);
}

// The segment plan (njode_gen_seg.h) serves the unmasked calls that ask for the loss and not for
// the path: training and loss-only evaluation.  NJODE_GEN_PLAN=lock keeps them on the lockstep
// plan (A/B measurements; tests compare the two plans).
bool use_seg(const Model& m, int n_obs, int call_flags) {
  // (the choice travels in the call flags, NJODE_C_GEN_LOCKSTEP: plan, forward and backward of one
  // step lay their buffers out by it and must agree -- the Python class reads NJODE_GEN_PLAN once
  // per step and sets the flag)
  const bool lock_only = (call_flags & NJODE_C_GEN_LOCKSTEP) != 0;
  // (use_rnn: the state after a jump depends on the state before it -- no independent items)
  return !lock_only && !m.a.masked && !m.a.rnn && n_obs > 0 && (call_flags & NJODE_C_GET_LOSS) &&
         !(call_flags & NJODE_C_RETURN_PATH);
}

// Lockstep plan: paths per tile of 16 chains.  A tile runs the three network evaluations of a jump
// for all its chains whenever ANY of its paths observes at that time, and a PhysioNet-shaped batch
// of 50 paths is 4 tiles on a 256-CU chip: fewer paths per tile (the other chains idle) shorten
// the serial chain of every tile -- same reasoning as q4_paths_per_tile of the specialised masked
// kernels (njode_api.hip).  The smallest power of two that keeps the tile count within the CUs
// and the training records (one per Euler step / jump and TILE) within record_budget_bytes()
// (njode_gen_host.h: 16 GB or 1/8 of the device's memory; round 4 allowed 40 GB).  NJODE_GEN_PT: A/B.
static int gen_paths_per_tile(const Model& m, int B, int nt, int K, int call_flags) {
  static const int env = getenv("NJODE_GEN_PT") ? atoi(getenv("NJODE_GEN_PT")) : 0;
  if (env == 1 || env == 2 || env == 4 || env == 8 || env == 16) return env;
  int pt = 1;
  while (pt < 16 && (B + pt - 1) / pt > 256) pt *= 2;
  if (call_flags & NJODE_C_SAVE_BWD) {
    auto bytes = [&](int p) {
      const double T = (double)((B + p - 1) / p);
      return T * ((double)(K > 0 ? K : 1) * m.a.ode.rec_rows +
                  (double)(nt > 0 ? nt : 1) * (m.a.enc.rec_rows + 2.0 * m.a.dec.rec_rows +
                                               (m.a.rnn ? m.a.gru.rec_rows : 0))) * 64.0;
    };
    const double budget = record_budget_bytes();
    while (pt < 16 && bytes(pt) > budget) pt *= 2;
  }
  return pt;
}

Layout make_layout(const Model& m, int B, int n_obs, int nt, int K, int call_flags) {
  Layout L;
  L.seg = use_seg(m, n_obs, call_flags);
  // (the segment plan's path tiles -- start values, tails -- are tiles of 16)
  L.PT = L.seg ? 16 : gen_paths_per_tile(m, B, nt, K, call_flags);
  L.T = cdiv(B, L.PT);
  L.NT = L.seg ? cdiv(n_obs, 16) : 0;
  const size_t T = (size_t)L.T, ntl = (size_t)(nt > 0 ? nt : 1);
  const size_t nr = (size_t)(n_obs > 0 ? n_obs : 1), nb = (size_t)B;
  L.sched = L.take(((size_t)2 * K + 3 * (size_t)nt + 1) * 4 + 64);
  L.jlo = L.take(((size_t)K + 2) * 4);
  L.dense = L.take(ntl * (size_t)B * 4);
  L.bad = L.take(256);
  if (L.seg) {
    L.t_of_row = L.take(nr * 4);
    L.item_prev = L.take(nr * 4);
    L.item_next = L.take(nr * 4);
    L.item_kbeg = L.take(nr * 4);
    L.item_len = L.take(nr * 4);
    L.key = L.take(nr * 4);
    L.key_sorted = L.take(nr * 4);
    L.iota = L.take(nr * 4);
    L.order = L.take(nr * 4);
    L.first_row = L.take(nb * 4);
    L.last_row = L.take(nb * 4);
    L.tail_key = L.take(nb * 4);
    L.tail_key_sorted = L.take(nb * 4);
    L.iota_b = L.take(nb * 4);
    L.tail_order = L.take(nb * 4);
    L.tile_base = L.take(((size_t)L.NT + 1) * 4);
    size_t b0 = 0, b1 = 0;
    (void)sort_pairs(nullptr, b0, nullptr, nullptr, nullptr, nullptr, n_obs, bits_for((unsigned)K), (hipStream_t)0);
    (void)sort_pairs(nullptr, b1, nullptr, nullptr, nullptr, nullptr, B, bits_for((unsigned)K), (hipStream_t)0);
    L.sort_tmp_bytes = b0 > b1 ? b0 : b1;
    L.sort_tmp = L.take(L.sort_tmp_bytes);
  }
  L.plan_end = L.total;
  L.frag = L.take((size_t)m.frag_floats * 4);
  L.loss_terms = L.take((nr > nb ? nr : nb) * 4);
  if (L.seg) {
    const size_t H = (size_t)m.a.H;
    L.h0row = L.take(nr * H * 4);
    L.h0start = L.take(nb * H * 4);
    L.h_end = L.take(nr * H * 4);
    if (call_flags & NJODE_C_SAVE_BWD) {
      L.lam_end = L.take(nr * H * 4);
      L.g_h0 = L.take(nr * H * 4);
      L.lam_start = L.take(nr * H * 4);
    }
  }
  if (call_flags & NJODE_C_SAVE_BWD) {
    L.slab = L.take((size_t)m.S * m.P * 4);
    // (segment plan: sum over the item tiles of their longest item <= K + B K / 16 records)
    L.rec_ode = L.take(((size_t)(K > 0 ? K : 1) * (T + 1) + 1) * m.a.ode.rec_rows * 64);
    if (L.seg) {
      // one encoder record per row tile and path tile, two readout records per row tile
      L.rec_enc = L.take(((size_t)L.NT + T) * m.a.enc.rec_rows * 64);
      L.rec_dec = L.take((size_t)L.NT * 2 * m.a.dec.rec_rows * 64);
      L.ybuf = L.flags = 0;
    } else {
      L.rec_enc = L.take(((size_t)nt * T + T) * m.a.enc.rec_rows * 64);
      L.rec_dec = L.take(ntl * T * 2 * m.a.dec.rec_rows * 64);
      L.ybuf = L.take(ntl * T * 2 * m.a.DO * 64);
      L.flags = L.take(ntl * T * 4);
      if (m.a.rnn) L.rec_gru = L.take(ntl * T * m.a.gru.rec_rows * 64);
    }
  } else {
    L.slab = L.rec_ode = L.rec_enc = L.rec_dec = L.ybuf = L.flags = 0;
  }
  return L;
}

int check_sizes(const NjodeBatch* b, const NjodeSchedule* s) {
  if (!b || !s) return fail(NJODE_E_BADARG, "null argument");
  if (b->batch_size <= 0 || b->n_obs < 0 || s->n_steps < 0 || s->n_times < 0)
    return fail(NJODE_E_BADARG, "negative size");
  if (!b->start_X || (b->n_obs > 0 && (!b->X || !b->obs_idx)))
    return fail(NJODE_E_BADARG, "null batch array");
  if ((s->n_steps > 0 && (!s->step_dt || !s->step_t)) ||
      (s->n_times > 0 && (!s->k_jump || !s->time_f32)) || !s->time_ptr)
    return fail(NJODE_E_BADARG, "null schedule array");
  if ((size_t)s->n_times * (size_t)b->batch_size > ((size_t)1 << 31))
    return fail(NJODE_E_UNSUPPORTED, "n_times x batch_size exceeds 2^31 cells");
  // k_jump (host array): non-decreasing and within [0, n_steps].  The segment plan sizes its
  // record buffers by it (a path's item lengths add up to at most n_steps only then), so a
  // malformed schedule must stop here, not write past the workspace.
  for (int i = 0, prev = 0; i < s->n_times; ++i) {
    const int k = s->k_jump[i];
    if (k < prev || k > s->n_steps) return fail(NJODE_E_BADARG, "k_jump is not non-decreasing within [0, n_steps]");
    prev = k;
  }
  return NJODE_OK;
}

// schedule copy + plan into the prefix at `pw`
int build_plan(const Layout& L, char* pw, const NjodeBatch* b, const NjodeSchedule* s, hipStream_t st) {
  const int K = s->n_steps, nt = s->n_times, B = b->batch_size, n = b->n_obs;
  char* dst = pw + L.sched;
  const size_t bdt = (size_t)K * 4, bt = (size_t)nt * 4, bp = ((size_t)nt + 1) * 4;
  const char* h0 = (const char*)s->step_dt;
  const bool contiguous = K > 0 && nt > 0 && (const char*)s->step_t == h0 + bdt &&
                          (const char*)s->k_jump == h0 + 2 * bdt &&
                          (const char*)s->time_f32 == h0 + 2 * bdt + bt &&
                          (const char*)s->time_ptr == h0 + 2 * bdt + 2 * bt;
  if (contiguous) {
    HIP_TRY(hipMemcpyAsync(dst, h0, 2 * bdt + 2 * bt + bp, hipMemcpyHostToDevice, st));
  } else {
    if (K > 0) {
      HIP_TRY(hipMemcpyAsync(dst, s->step_dt, bdt, hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(dst + bdt, s->step_t, bdt, hipMemcpyHostToDevice, st));
    }
    if (nt > 0) {
      HIP_TRY(hipMemcpyAsync(dst + 2 * bdt, s->k_jump, bt, hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(dst + 2 * bdt + bt, s->time_f32, bt, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipMemcpyAsync(dst + 2 * bdt + 2 * bt, s->time_ptr, bp, hipMemcpyHostToDevice, st));
  }
  const int* k_jump = (const int*)(dst + 2 * bdt);
  const int* time_ptr = (const int*)(dst + 2 * bdt + 2 * bt);
  int* dense = (int*)(pw + L.dense);
  int* bad = (int*)(pw + L.bad);
  HIP_TRY(hipMemsetAsync(bad, 0, 4, st));
  if (nt > 0) HIP_TRY(hipMemsetAsync(dense, 0xFF, (size_t)nt * B * 4, st));
  if (n > 0) k_gen_rows<<<cdiv(n, 256), 256, 0, st>>>(time_ptr, nt, n, b->obs_idx, B, dense, bad);
  k_gen_jlo<<<cdiv(K + 2, 256), 256, 0, st>>>(k_jump, nt, K, (int*)(pw + L.jlo));
  if (L.seg) {
    // items = rows linked along their path, sorted by length; tails sorted by length
    k_gseg_init<<<cdiv(n, 256), 256, 0, st>>>(n, K, (int*)(pw + L.iota), (int*)(pw + L.t_of_row),
                                              (int*)(pw + L.item_prev), (int*)(pw + L.item_next),
                                              (int*)(pw + L.item_kbeg), (int*)(pw + L.item_len),
                                              (unsigned*)(pw + L.key));
    k_gseg_link<<<cdiv(B, 64), 64, 0, st>>>(
        B, nt, K, dense, k_jump, (int*)(pw + L.t_of_row), (int*)(pw + L.item_prev), (int*)(pw + L.item_next),
        (int*)(pw + L.item_kbeg), (int*)(pw + L.item_len), (unsigned*)(pw + L.key), (int*)(pw + L.first_row),
        (int*)(pw + L.last_row), (unsigned*)(pw + L.tail_key), (int*)(pw + L.iota_b));
    size_t bytes = L.sort_tmp_bytes;
    HIP_TRY(sort_pairs(pw + L.sort_tmp, bytes, (const unsigned*)(pw + L.key), (unsigned*)(pw + L.key_sorted),
                       (const int*)(pw + L.iota), (int*)(pw + L.order), n, bits_for((unsigned)K), st));
    bytes = L.sort_tmp_bytes;
    HIP_TRY(sort_pairs(pw + L.sort_tmp, bytes, (const unsigned*)(pw + L.tail_key),
                       (unsigned*)(pw + L.tail_key_sorted), (const int*)(pw + L.iota_b),
                       (int*)(pw + L.tail_order), B, bits_for((unsigned)K), st));
    k_gseg_tiles<<<1, 1024, 0, st>>>((const int*)(pw + L.order), (const int*)(pw + L.item_len), n, L.NT,
                                     (int*)(pw + L.tile_base));
  }
  static const bool validate = getenv("NJODE_VALIDATE") && atoi(getenv("NJODE_VALIDATE")) != 0;
  if (validate) {
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h) return fail(NJODE_E_BADARG, "obs_idx has entries outside [0, batch_size)");
  }
  HIP_TRY(hipGetLastError());
  return NJODE_OK;
}

struct Call {
  Model m;
  Layout L;
  GArgs a;
  GSeg g;
};

int prepare(Call& c, const NjodeDims* dims, const float* params, const NjodeBatch* b,
            const NjodeSchedule* s, int call_flags, float weight, float dropout_p, uint64_t seed,
            void* ws, size_t ws_bytes) {
  const char* why;
  if (!build_model(dims, c.m, &why)) return fail(NJODE_E_UNSUPPORTED, "generic kernels: %s", why);
  int rc = check_sizes(b, s);
  if (rc) return rc;
  if (!params || !ws) return fail(NJODE_E_BADARG, "null argument");
  const int B = b->batch_size, n_obs = b->n_obs, K = s->n_steps, nt = s->n_times;
  const bool masked = (dims->flags & NJODE_F_MASKED) != 0;
  if (masked && n_obs > 0 && !b->M) return fail(NJODE_E_BADARG, "masked model needs M");
  if ((call_flags & NJODE_C_GET_LOSS) && !b->n_obs_ot) return fail(NJODE_E_BADARG, "get_loss needs n_obs_ot");
  if (dropout_p < 0.0f || dropout_p >= 1.0f) return fail(NJODE_E_BADARG, "dropout_p");
  c.L = make_layout(c.m, B, n_obs, nt, K, call_flags);
  if (ws_bytes < c.L.total)
    return fail(NJODE_E_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, c.L.total);
  const bool planned = (call_flags & NJODE_C_PLAN_READY) != 0;
  if (planned && !b->plan) return fail(NJODE_E_BADARG, "NJODE_C_PLAN_READY without NjodeBatch.plan");
  char* w = (char*)ws;
  char* pw = planned ? (char*)b->plan : w;
  c.a = c.m.a;
  GArgs& a = c.a;
  a.P = params;
  a.frag = (const float*)(w + c.L.frag);
  a.B = B; a.T = c.L.T; a.PT = c.L.PT; a.n_obs = n_obs; a.K = K; a.n_times = nt;
  a.start_X = b->start_X; a.X = b->X; a.M = b->M; a.n_obs_ot = b->n_obs_ot;
  a.inv_batch = 1.0f / b->loss_batch_size;
  a.g_hT = nullptr;      // (gen_backward sets it)
  a.gid0 = (unsigned long long)b->path_id_offset;
  const float* sb = (const float*)(pw + c.L.sched);
  a.step_dt = sb;
  a.step_t = sb + K;
  a.time_f32 = sb + 2 * (size_t)K + nt;
  a.jlo = (const int*)(pw + c.L.jlo);
  a.dense = (const int*)(pw + c.L.dense);
  a.loss_terms = (float*)(w + c.L.loss_terms);
  a.save = (call_flags & NJODE_C_SAVE_BWD) ? 1 : 0;
  if (a.save) {
    a.rec_ode = (float*)(w + c.L.rec_ode);
    a.rec_enc = (float*)(w + c.L.rec_enc);
    a.rec_dec = (float*)(w + c.L.rec_dec);
    a.ybuf = (float*)(w + c.L.ybuf);
    a.flags = (int*)(w + c.L.flags);
    a.rec_gru = a.rnn ? (float*)(w + c.L.rec_gru) : nullptr;
  }
  if (c.m.a.D != c.m.a.DO && (call_flags & (NJODE_C_GET_LOSS | NJODE_C_SAVE_BWD)))
    return fail(NJODE_E_BADARG, "the loss compares X with the readout: a model with input_size != output_size "
                                "runs prediction calls only (get_loss = False)");
  a.want_loss = (call_flags & NJODE_C_GET_LOSS) ? 1 : 0;
  a.want_path = (call_flags & NJODE_C_RETURN_PATH) ? 1 : 0;
  const bool any_hidden = c.m.a.ode.nl > 1 || c.m.a.enc.nl > 1 || c.m.a.dec.nl > 1;
  a.drop = ((call_flags & NJODE_C_TRAIN) && dropout_p > 0.0f && any_hidden) ? 1 : 0;
  unsigned thr = (unsigned)(dropout_p * 65536.0f + 0.5f);
  if (thr > 65535u) thr = 65535u;
  a.dc.seed_lo = (uint32_t)seed;
  a.dc.seed_hi = (uint32_t)(seed >> 32);
  a.dc.thr16 = a.drop ? thr : 0;
  a.keep = 1.0f - (float)a.dc.thr16 / 65536.0f;
  a.dc.inv_keep = 1.0f / a.keep;
  a.weight = weight;
#ifdef NJ_GEN_ABL
  a.dbg = getenv("NJODE_GEN_DBG") ? atoi(getenv("NJODE_GEN_DBG")) : 0;
#endif
  memset(&c.g, 0, sizeof(c.g));
  if (c.L.seg) {
    // Waves per workgroup: one wave per output tile of the widest layer fills a CU with a single
    // tile and is the fastest way through ONE tile; when the grid holds many times more tiles than
    // fit on the chip at once, four waves per tile (each walks several output tiles) run four
    // tiles per CU and pay the per-wave, per-layer scalar work a quarter as often
    // (profiles/r03_generic_waves_per_tile.txt: width 100, 20 000 paths: 8.0 -> 5.8 ms).
    // (per device: a process may drive GPUs of different sizes)
    static std::mutex cu_mu;
    static int cu_of_dev[64] = {};
    int n_cu = 256;
    {
      int dev = 0;
      if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        std::lock_guard<std::mutex> lk(cu_mu);
        if (cu_of_dev[dev] <= 0) {
          int v = 0;
          if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
          cu_of_dev[dev] = v;
        }
        n_cu = cu_of_dev[dev];
      }
    }
    if (!getenv("NJODE_GEN_NW") && c.m.nw > 4 && (long long)c.L.NT >= 16LL * n_cu) c.m.nw = 4;
  }
  {
    // output tiles per wave, now that the waves per workgroup are known
    GNet* nets[3] = {&a.ode, &a.enc, &a.dec};
    for (GNet* N : nets)
      for (int l = 0; l < N->nl; ++l) {
        N->l[l].per = cdiv(N->l[l].MT, c.m.nw);
        N->l[l].pert = cdiv(N->l[l].MTT, c.m.nw);
      }
    a.gru.l[0].per = cdiv(a.gru.l[0].MT, c.m.nw);
    a.gru.l[0].pert = cdiv(a.gru.l[0].MTT, c.m.nw);
  }
  if (c.L.seg) {
    GSeg& g = c.g;
    g.order = (const int*)(pw + c.L.order);
    g.item_prev = (const int*)(pw + c.L.item_prev);
    g.item_next = (const int*)(pw + c.L.item_next);
    g.item_kbeg = (const int*)(pw + c.L.item_kbeg);
    g.item_len = (const int*)(pw + c.L.item_len);
    g.t_of_row = (const int*)(pw + c.L.t_of_row);
    g.first_row = (const int*)(pw + c.L.first_row);
    g.last_row = (const int*)(pw + c.L.last_row);
    g.tile_base = (const int*)(pw + c.L.tile_base);
    g.tail_order = (const int*)(pw + c.L.tail_order);
    g.k_jump = (const int*)(sb + 2 * (size_t)K);
    g.obs_idx = b->obs_idx;
    g.NT = c.L.NT;
    g.TB = c.L.T;
    g.h0row = (float*)(w + c.L.h0row);
    g.h0start = (float*)(w + c.L.h0start);
    g.h_end = (float*)(w + c.L.h_end);
    if (a.save) {
      g.lam_end = (float*)(w + c.L.lam_end);
      g.g_h0 = (float*)(w + c.L.g_h0);
      g.lam_start = (float*)(w + c.L.lam_start);
    }
    g.loss_rows = a.loss_terms;
  }
  return NJODE_OK;
}

int set_lds(const void* fn, int bytes) {
  if (bytes > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return NJODE_OK;
}

}  // namespace

namespace njode {
namespace gen {

bool gen_supported(const NjodeDims* dims) {
  Model m;
  const char* why;
  return build_model(dims, m, &why);
}

size_t gen_param_count(const NjodeDims* dims) {
  Model m;
  const char* why;
  return build_model(dims, m, &why) ? (size_t)m.P : 0;
}

int gen_workspace_bytes(const NjodeDims* dims, int B, int n_obs, int nt, int K, int call_flags,
                        size_t* out, bool plan_only) {
  Model m;
  const char* why;
  if (!build_model(dims, m, &why)) return fail(NJODE_E_UNSUPPORTED, "generic kernels: %s", why);
  if (!out || B <= 0 || n_obs < 0 || nt < 0 || K < 0) return fail(NJODE_E_BADARG, "bad size");
  const Layout L = make_layout(m, B, n_obs, nt, K, call_flags);
  *out = plan_only ? L.plan_end : L.total;
  return NJODE_OK;
}

int gen_plan(const NjodeDims* dims, const NjodeBatch* b, const NjodeSchedule* s, int call_flags,
             void* plan, size_t plan_bytes, hipStream_t st) {
  Model m;
  const char* why;
  if (!build_model(dims, m, &why)) return fail(NJODE_E_UNSUPPORTED, "generic kernels: %s", why);
  int rc = check_sizes(b, s);
  if (rc) return rc;
  if (!plan) return fail(NJODE_E_BADARG, "null plan buffer");
  const Layout L = make_layout(m, b->batch_size, b->n_obs, s->n_times, s->n_steps, call_flags);
  if (plan_bytes < L.plan_end) return fail(NJODE_E_WORKSPACE, "plan buffer too small: %zu < %zu", plan_bytes, L.plan_end);
  return build_plan(L, (char*)plan, b, s, st);
}

int gen_forward(const NjodeDims* dims, const float* params, const NjodeBatch* b,
                const NjodeSchedule* s, int call_flags, float weight, float dropout_p,
                uint64_t seed, float* hT, float* loss, float* path_h, float* path_y, void* ws,
                size_t ws_bytes, hipStream_t st) {
  Call c;
  int rc = prepare(c, dims, params, b, s, call_flags, weight, dropout_p, seed, ws, ws_bytes);
  if (rc) return rc;
  const bool want_loss = call_flags & NJODE_C_GET_LOSS, want_path = call_flags & NJODE_C_RETURN_PATH;
  if ((want_loss && !loss) || (want_path && (!path_h || !path_y))) return fail(NJODE_E_BADARG, "null output");
  if ((call_flags & NJODE_C_SAVE_BWD) && !want_loss) return fail(NJODE_E_BADARG, "NJODE_C_SAVE_BWD needs NJODE_C_GET_LOSS");
  c.a.hT = hT;
  c.a.path_h = path_h;
  c.a.path_y = path_y;
  if (!(call_flags & NJODE_C_PLAN_READY)) {
    if ((rc = build_plan(c.L, (char*)ws, b, s, st))) return rc;
  }
  {
    Prof ps("k_gen_pack", st);
    k_gen_pack<<<cdiv(c.m.pack.total, 256), 256, 0, st>>>(params, (float*)c.a.frag, c.m.pack);
  }
  if (c.L.seg) {
    const int nth = c.m.nw * 64, lds = c.m.lds_bytes;
    c.g.tails = hT != nullptr;
    if ((rc = set_lds((const void*)k_gseg_enc, lds)) || (rc = set_lds((const void*)k_gseg_ode_fwd, lds)) ||
        (rc = set_lds((const void*)k_gseg_mid, lds)))
      return rc;
    {
      Prof ps("k_gseg_enc", st);
      k_gseg_enc<<<c.g.NT + c.g.TB, nth, lds, st>>>(c.a, c.g);
    }
    {
      Prof ps("k_gseg_ode_fwd", st);
      k_gseg_ode_fwd<<<c.g.NT + (c.g.tails ? c.g.TB : 0), nth, lds, st>>>(c.a, c.g);
    }
    {
      Prof ps("k_gseg_mid", st);
      k_gseg_mid<<<c.g.NT, nth, lds, st>>>(c.a, c.g);
    }
    k_gen_sum<<<1, 1024, 0, st>>>(c.a.loss_terms, b->n_obs, loss);
    HIP_TRY(hipGetLastError());
    return NJODE_OK;
  }
  if (c.a.save) HIP_TRY(hipMemsetAsync(c.a.flags, 0, (size_t)(s->n_times > 0 ? s->n_times : 1) * c.L.T * 4, st));
  if ((rc = set_lds((const void*)k_gen_fwd, c.m.lds_bytes))) return rc;
  {
    Prof ps("k_gen_fwd", st);
    k_gen_fwd<<<c.L.T, c.m.nw * 64, c.m.lds_bytes, st>>>(c.a);
  }
  if (want_loss) k_gen_sum<<<1, 1024, 0, st>>>(c.a.loss_terms, b->batch_size, loss);
  HIP_TRY(hipGetLastError());
  return NJODE_OK;
}

int gen_backward(const NjodeDims* dims, const float* params, const NjodeBatch* b,
                 const NjodeSchedule* s, int call_flags, float weight, float dropout_p,
                 uint64_t seed, const float* grad_loss, float* grad_params, void* ws,
                 size_t ws_bytes, hipStream_t st) {
  Call c;
  int rc = prepare(c, dims, params, b, s, call_flags, weight, dropout_p, seed, ws, ws_bytes);
  if (rc) return rc;
  if (!grad_loss || !grad_params) return fail(NJODE_E_BADARG, "null gradient pointer");
  if (!(call_flags & NJODE_C_SAVE_BWD) || !(call_flags & NJODE_C_GET_LOSS))
    return fail(NJODE_E_BADARG, "backward needs a forward with NJODE_C_SAVE_BWD | NJODE_C_GET_LOSS");
  // (schedule copy, plan, fragment tables and records are where the forward left them)
  if (b->grad_hT && c.L.seg)
    return fail(NJODE_E_UNSUPPORTED, "NjodeBatch.grad_hT: the segment plan does not differentiate through hT; "
                                     "run the step with NJODE_C_GEN_LOCKSTEP");
  c.a.g_hT = b->grad_hT;
  if (c.L.seg) {
    const int nth = c.m.nw * 64, lds = c.m.lds_bytes;
    if ((rc = set_lds((const void*)k_gseg_ode_bwd, lds)) || (rc = set_lds((const void*)k_gseg_enc_bwd, lds)))
      return rc;
    {
      Prof ps("k_gseg_ode_bwd", st);
      k_gseg_ode_bwd<<<c.g.NT, nth, lds, st>>>(c.a, c.g);
    }
    {
      Prof ps("k_gseg_enc_bwd", st);
      k_gseg_enc_bwd<<<c.g.NT + c.g.TB, nth, lds, st>>>(c.a, c.g);
    }
  } else {
    if ((rc = set_lds((const void*)k_gen_bwd, c.m.lds_bytes))) return rc;
    Prof ps("k_gen_bwd", st);
    k_gen_bwd<<<c.L.T, c.m.nw * 64, c.m.lds_bytes, st>>>(c.a);
  }
  float* slab = (float*)((char*)ws + c.L.slab);
  const GArgs& a = c.a;
  const long long T = c.L.T, nt = a.n_times;
  const int* n_rec_dev = nullptr;
  auto dw = [&](const auto& N, const float* rec, long long n_rec, const int* flags, int flag_div,
                long long n_flagged) {
    for (int l = 0; l < N.nl; ++l) {
      const GLayer& Ly = N.l[l];
      GDw d;
      d.rec = rec;
      d.n_rec = n_rec;
      d.n_rec_dev = n_rec_dev;
      d.rec_floats = N.rec_rows * 16;
      d.flags = flags;
      d.flag_div = flag_div;
      d.n_flagged = n_flagged;
      d.a_row = Ly.a_row;
      d.d_row = Ly.d_row;
      d.n_in = Ly.n_in;
      d.n_out = Ly.n_out;
      d.w_off = Ly.w_off;
      d.b_off = Ly.b_off;
      d.kind = Ly.kind;
      d.P = c.m.P;
      d.tiles_m = Ly.MT;
      d.tiles_n = cdiv(Ly.n_in + 1, 16);
      d.slab = slab;
      const int tb = cdiv(d.tiles_m, DW_TM) * cdiv(d.tiles_n, DW_TN);
      k_gen_dw<<<dim3(tb, c.m.S), 256, 0, st>>>(d);
    }
  };
  if (c.L.seg) {
    Prof ps("k_gen_dw", st);
    n_rec_dev = c.g.tile_base + c.g.NT;        // the number of ODE records is known on the device only
    dw(a.ode, a.rec_ode, (long long)a.K * (T + 1) + 1, nullptr, 1, 0);
    n_rec_dev = nullptr;
    dw(a.enc, a.rec_enc, (long long)c.g.NT + c.g.TB, nullptr, 1, 0);
    dw(a.dec, a.rec_dec, (long long)c.g.NT * 2, nullptr, 1, 0);
  } else {
    Prof ps("k_gen_dw", st);
    dw(a.ode, a.rec_ode, (long long)a.K * T, nullptr, 1, 0);
    if (a.rnn) {
      // (the encoder only produced the start states; the jumps went through the GRU cell)
      dw(a.enc, a.rec_enc + (size_t)nt * T * a.enc.rec_rows * 16, T, nullptr, 1, 0);
      dw(a.gru, a.rec_gru, nt * T, a.flags, 1, nt * T);
    } else {
      dw(a.enc, a.rec_enc, nt * T + T, a.flags, 1, nt * T);
    }
    dw(a.dec, a.rec_dec, nt * T * 2, a.flags, 2, nt * T * 2);
  }
  k_gen_reduce<<<cdiv(c.m.P, 256), 256, 0, st>>>(slab, c.m.S, c.m.P, grad_loss, grad_params);
  HIP_TRY(hipGetLastError());
  return NJODE_OK;
}

}  // namespace gen
}  // namespace njode

#ifdef NJ_GEN_STAMPS
// diagnostic build only: read and clear the phase stamps of njode_gen.h
extern "C" int njode_gen_debug_stamps(unsigned long long* out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(njode::gen::g_gen_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long z[16] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(njode::gen::g_gen_stamps), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif
