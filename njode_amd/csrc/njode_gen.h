// njode_gen.h -- shape-generic NJ-ODE kernels on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// The specialised kernels (njode_mfma*.h, njode_ode2.h) keep a layer's units in registers and
// a network's A-fragments in VGPRs / LDS, so every model shape is a template instantiation and
// a layer wider than 63 units does not fit.  This family takes the shape at RUN TIME:
// any input / hidden / output size, per network any number of hidden layers (<= GEN_MAXL - 1 = NJODE_MAX_HIDDEN)
// of any widths with tanh or relu each, the three networks independent of each other, masked or
// not, every residual case, dropout, and (round 4) the GRU jump of use_rnn models (gru_* below).  It serves every shape the build table has no
// specialisation for (reference grids: widths 80 ... 400, nn_desc = None with hidden_size 50 /
// 100, the climate shape d = 5: NJODE/parallel_train.py:304-305, 366-371, 433-470, 609, 650,
// 712).  Masked models, return_path and get_loss = False run the lockstep plan of this file (one
// chain per path over the shared grid, models.py:430-511); unmasked loss calls run the segment
// plan of njode_gen_seg.h on the same building blocks.
//
// Execution model
//   * one workgroup (NW waves) advances a TILE of 16 chains; vectors over the tile live in LDS
//     images (layout: pix() below);
//   * a layer is out = W in: the A operand (16 output units x 4 input units) streams from a
//     packed fragment table in global memory (L2-resident: every workgroup reads the same
//     table) through a ring of registers, the B operands (4 input units x 16 chains) of four
//     k-steps are one ds_read_b128 of the input image, wave w owns the output tiles
//     [w per, (w + 1) per); bias = the constant-1 row behind the inputs; the epilogue applies
//     activation + dropout and writes the output image;
//   * training calls store every layer input of every network evaluation (one record per
//     (Euler step | jump time, tile)); the adjoint sweep walks the events in reverse with the
//     transposed fragment tables and stores the delta of every layer output next to it;
//   * weight gradients are then plain GEMMs over those records, dW = sum_records delta x input,
//     K = the 16 chains of a record: one float4 load per operand tile carries all four k-steps
//     (k_gen_dw); slabs + fixed-order reduction, no float atomics.
#pragma once
#include "njode_device.h"

namespace njode {
namespace gen {

typedef float f32x4 __attribute__((ext_vector_type(4)));
NJ_DEV f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int GEN_MAXL = NJODE_MAX_HIDDEN + 1;   // layers per network (hidden layers + the output layer)
constexpr int QU = 4;            // k-steps are padded to a multiple of QU (prefetch depth)
constexpr uint32_t G_NET_ODE = 0, G_NET_ENC = 1, G_NET_DEC = 2, G_NET_DEC_BJ = 3, G_NET_DEC_ROW = 4;
constexpr uint32_t G_TKEY_START = 0xffffffffu;

struct GLayer {
  int n_in, n_out;      // units in / out (bias not counted)
  int act;              // activation of the OUTPUT (ACT_TANH / ACT_RELU); -1: linear (last layer)
  int w_off, b_off;     // offsets of W [out][in] and b [out] in the flat parameter vector
  int f_off, ft_off;    // fragment tables (floats from the table base): forward, transposed
  int Qp, MT;           // forward: k-steps (inputs + bias, padded to QU), output tiles of 16
  int QTp, MTT;         // transposed: k-steps over the outputs (padded), tiles over the inputs
  int a_row, d_row;     // record rows: this layer's INPUT vector / the delta of its OUTPUT
  int per, pert;        // output tiles per wave of the launch: forward / transposed product
  int kind;             // 0: a Linear layer; 1: the GRU cell's two Linear maps as ONE layer (gru_w)
};
struct GNet {
  int nl;               // layers = hidden layers + 1
  int rec_rows;         // rows (of 16 floats) of one evaluation record
  int n_in, n_out;
  GLayer l[GEN_MAXL];
};
struct GNet1 {          // a one-layer "network" (same members: net_forward / net_backward take either)
  int nl, rec_rows, n_in, n_out;
  GLayer l[1];
};

// ---- GRU jump (use_rnn; reference models.py:202-217: h[i_obs] = GRUCell(tanh(X_obs), tanh(h[i_obs]))) ----
// torch.nn.GRUCell:  r = sig(W_ir x + b_ir + W_hr h + b_hr),  z = sig(W_iz x + b_iz + W_hz h + b_hz),
//                    n = tanh(W_in x + b_in + r (W_hn h + b_hn)),  h' = (1 - z) n + z h.
// Here the two Linear maps are ONE layer of the family's machinery: input [x (D); h (H); 1], output
//   rows [0, H): r's pre-activation   [H, 2H): z's   [2H, 3H): W_in x + b_in   [3H, 4H): W_hn h + b_hn
// -- a block matrix with zeros where the cell has no weight (a jump is rare next to the Euler
// steps; the zeros cost nothing that matters) -- so fragment tables, records, the transposed
// product and the weight-gradient GEMM are the ones every other layer uses; only the value of
// entry (o, i) (gru_w) and where its gradient goes (gru_dw_store) are the cell's own.
// Parameters at w_off, state_dict order: weight_ih [3H][D], weight_hh [3H][H], bias_ih [3H], bias_hh [3H].
NJ_DEV float gru_w(const float* __restrict__ P, const GLayer& L, int o, int i) {
  const int H = L.n_out >> 2, D = L.n_in - H;
  const float *Wih = P + L.w_off, *Whh = Wih + 3 * H * D, *bih = Whh + 3 * H * H, *bhh = bih + 3 * H;
  const int gate = o / H, j = o - gate * H;
  if (gate < 2) return i < D ? Wih[o * D + i] : (i < D + H ? Whh[o * H + (i - D)] : bih[o] + bhh[o]);
  if (gate == 2) return i < D ? Wih[o * D + i] : (i < D + H ? 0.0f : bih[o]);
  const int r = 2 * H + j;
  return i < D ? 0.0f : (i < D + H ? Whh[r * H + (i - D)] : bhh[r]);
}
// gradient of entry (o, i) of that layer -> the cell's parameters in a slab row (i == n_in: bias)
NJ_DEV void gru_dw_store(float* __restrict__ row, int w_off, int n_in, int n_out, int o, int i, float v) {
  const int H = n_out >> 2, D = n_in - H;
  float *Wih = row + w_off, *Whh = Wih + 3 * H * D, *bih = Whh + 3 * H * H, *bhh = bih + 3 * H;
  const int gate = o / H, j = o - gate * H;
  if (i < D) {
    if (gate < 3) Wih[o * D + i] = v;
  } else if (i < D + H) {
    if (gate < 2) Whh[o * H + (i - D)] = v;
    else if (gate == 3) Whh[(2 * H + j) * H + (i - D)] = v;
  } else {
    if (gate < 2) { bih[o] = v; bhh[o] = v; }
    else if (gate == 2) bih[o] = v;
    else bhh[2 * H + j] = v;
  }
}

struct GArgs {
  const float* P;
  const float* frag;
  GNet ode, enc, dec;
  GNet1 gru;             // use_rnn: the GRU cell as one layer (above)
  int rnn;
  float* rec_gru;        // [n_times][T] records of the GRU jumps
  int D, H, DO, IN0;
  int masked, curt, enc_case, enc_mult, dec_case, dec_mult, loss_easy;
  int B, T, n_obs, K, n_times;
  int PT;                // lockstep kernels: paths per tile of 16 chains (njode_gen.hip, gen_paths_per_tile)
  const float* start_X;
  const float* X;
  const float* M;
  const int* n_obs_ot;
  float inv_batch;
  unsigned long long gid0;
  const float* step_dt;
  const float* step_t;
  const float* time_f32;
  const int* jlo;        // [K + 2] first time index whose jump happens at Euler step >= k
  const int* dense;      // [n_times][B] row of path b at time i, or -1
  float* rec_ode;        // [K][T] records
  float* rec_enc;        // [n_times][T] jump records, then [T] start records
  float* rec_dec;        // [n_times][T][2]: readout before the jump, readout after it
  float* ybuf;           // [n_times][T][2][DO][16]: y_bj, y
  int* flags;            // [n_times][T] 1: the tile had an observation at this time
  float* hT;
  float* path_h;
  float* path_y;
  float* loss_terms;     // [B]
  const float* g_hT;     // [B][H] upstream gradient of hT for the lockstep backward (NjodeBatch.grad_hT), or null
  int save, want_path, want_loss, drop;
  DropCtx dc;
  float keep, weight;
  int img_rows;          // rows of one ping-pong image
  int dbg;               // ablation bits of the diagnostic build (-DNJ_GEN_ABL), 0 otherwise
};
#ifdef NJ_GEN_ABL
#define ABL(dbg, bit) ((dbg) & (bit))
#else
#define ABL(dbg, bit) 0
#endif

// ---- LDS carve-up (floats), identical in the forward and the sweep --------------------------
struct GLds {
  lfp img0, img1, h, tx, y, ybj, xin, mk, hn, xr, misc;
  int* rows;
  NJ_DEV void carve(lfp base, const GArgs& a) {
    lfp p = base;
    const int mx = a.D > a.DO ? a.D : a.DO;
    img0 = p; p += a.img_rows * 16;
    img1 = p; p += a.img_rows * 16;
    h = p; p += a.H * 16;
    tx = p; p += a.D * 16;
    y = p; p += mx * 16;
    ybj = p; p += mx * 16;
    xin = p; p += mx * 16;
    mk = p; p += mx * 16;
    hn = p; p += a.H * 16;
    xr = p; p += mx * 16;
    misc = p; p += 8 * 16;     // tau, loss, scale, lam scratch ...
    rows = (int*)p;
  }
};
__host__ __device__ inline int gen_lds_floats(int img_rows, int D, int H, int DO) {
  const int mx = D > DO ? D : DO;
  return 2 * img_rows * 16 + 2 * H * 16 + D * 16 + 5 * mx * 16 + 8 * 16 + 16;
}

NJ_DEV float tanh_acc(float x) { return tanh_accurate(x); }   // (njode_device.h: <= ~1.3 ulp)
NJ_DEV float act_rt(int act, float z) { return act == ACT_TANH ? tanh_acc(z) : fmaxf(z, 0.0f); }
NJ_DEV float dact_rt(int act, float av) { return act == ACT_TANH ? 1.0f - av * av : (av > 0.0f ? 1.0f : 0.0f); }

// dropout: one hash per (call seed, global path, time key, network) and chain, one fmix per two units
NJ_DEV uint32_t drop_base(const DropCtx& dc, unsigned long long gid, uint32_t tkey, uint32_t net) {
  return drop_state(dc, (uint32_t)gid, (uint32_t)(gid >> 32), tkey, net);
}
// one word per PAIR of units: two 16-bit keep decisions (unit 2p: low half, unit 2p + 1: high half)
NJ_DEV uint32_t drop_word(uint32_t base, int layer, int pair) {
  return fmix32(base ^ ((uint32_t)layer * 0x9e3779b9u + (uint32_t)pair * 0x85ebca6bu + 0x632be5abu));
}

// ---- LDS images ------------------------------------------------------------------------------
// A vector over the tile (unit u, chain c) lives at pix(u * 16 + c): within every block of 16
// units the four k-steps a lane group needs next to each other are adjacent,
//   [block u / 16][lane group u % 4][chain][k-step (u / 4) % 4],
// so the B operands of FOUR MFMAs are one conflict-free ds_read_b128 (lane (g, c) reads the 16
// bytes at block * 256 + g * 64 + c * 4), and an accumulator register r of lane (g, c) -- unit
// 16 mt + 4 g + r -- goes to mt * 256 + r * 64 + c * 4 + g: 64 consecutive words per store.
NJ_DEV int pix(int e) {
  const int u = e >> 4, c = e & 15;
  return ((u >> 4) << 8) | ((u & 3) << 6) | (c << 2) | ((u >> 2) & 3);
}
NJ_DEV void img_put(lfp img, lfp lin, int rows) {     // image <- linear [unit][chain] vector
  for (int e = threadIdx.x; e < rows * 16; e += blockDim.x) img[pix(e)] = lin[e];
}
NJ_DEV void img_get(lfp lin, lfp img, int rows) {     // linear vector <- image
  for (int e = threadIdx.x; e < rows * 16; e += blockDim.x) lin[e] = img[pix(e)];
}
// rows [0, rows) of an image -> global record rows (linear [unit][16]): one ds_read_b128 and four
// coalesced 256-byte stores per 4 x 64 values
NJ_DEV void img_store(float* __restrict__ dst, lfp img, int rows) {
  const int nf4 = ((rows + 15) >> 4) << 6;
  for (int j = threadIdx.x; j < nf4; j += blockDim.x) {
    const f4 v = *(lf4p)(img + (j << 2));
    const int blk = j >> 6, gi = (j >> 4) & 3, c = j & 15;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int u = (blk << 4) + (qq << 2) + gi;
      if (u < rows) dst[u * 16 + c] = v[qq];
    }
  }
}

// ---- one matrix product of the tile ------------------------------------------------------------
// Fragment table of a layer: [output tile][quad of k-steps][64 lanes][4 k-steps] (k_gen_pack), so a
// wave's tiles are one contiguous stream of QUADS (16 bytes per lane, 1 KB per wave), read RING
// quads at a time.
constexpr int RING = 8;     // quads of a short row (no prefetch)
constexpr int LCH = 4;      // quads per chunk of a long row (two register sets)
typedef const f4 __attribute__((address_space(1))) * gf4p;      // global memory, 16 B
struct FragStream {          // the quads of one wave in one layer (all members wave-uniform)
  gf4p base;                 // quad n of the stream, lane l: base[n * 64 + l]
  int Qq;                    // quads per tile
  int ntl;                   // tiles of this wave
  int mt0;                   // its first tile
};
NJ_DEV int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
NJ_DEV FragStream frag_stream(const float* __restrict__ ft, int Qq, int MT, int per) {
  const int wv = wave_id();
  FragStream s;
  s.mt0 = wv * per;
  s.ntl = MT - s.mt0 < per ? MT - s.mt0 : per;
  if (s.ntl < 0) s.ntl = 0;
  s.Qq = Qq;
  s.base = (gf4p)(ft + (size_t)s.mt0 * Qq * 256);
  return s;
}
// all tiles of this wave.  pre(tile) right behind a tile's first fragment loads (loads of its own
// that the epilogue needs ride along), epi(tile, acc) once per tile.
template <class PRE, class EPI>
NJ_DEV void layer_product(const FragStream s, lfp in, PRE pre, EPI epi, int dbg = 0) {
  if (s.ntl <= 0) return;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  if (s.Qq <= RING) {
    // short rows (<= 8 quads = 128 input units): the whole tile's fragments at once, no ring --
    // a fraction of the ring's bookkeeping instructions, and the same one L2 round trip per tile
    lf4p bq = (lf4p)(in + (g << 6) + (c << 2));
    for (int t = 0; t < s.ntl; ++t) {
      gf4p tp = s.base + (size_t)t * s.Qq * 64 + lane;
      f4 fr[RING];
#pragma unroll
      for (int i = 0; i < RING; ++i)
        if (i < s.Qq) fr[i] = tp[i * 64];
      pre(s.mt0 + t);
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < RING; ++i)
        if (i < s.Qq) {
          const f4 bv = bq[i * 64];
          acc0 = mfma4(fr[i][0], bv[0], acc0);
          acc1 = mfma4(fr[i][1], bv[1], acc1);
          acc0 = mfma4(fr[i][2], bv[2], acc0);
          acc1 = mfma4(fr[i][3], bv[3], acc1);
        }
      epi(s.mt0 + t, acc0 + acc1);
    }
    return;
  }
  // long rows: chunks of LCH quads through two register sets (2 x 16 registers: the kernels run at
  // the 128-VGPR budget of 16-wave workgroups) -- the loads of chunk n + 1 are
  // issued before the MFMAs of chunk n, unconditionally and in program order (a tile's last
  // chunk reads up to LCH - 1 quads past the tile: the next tile's, the next table's or the
  // padding behind the last table; never used), so the compiler's vmcnt counting stays exact
  lf4p bp = (lf4p)(in + (g << 6) + (c << 2));
  const int NCH = (s.Qq + LCH - 1) / LCH;
  auto loadc = [&](f4 (&x)[LCH], gf4p tp, int ch) {
    gf4p p = tp + (size_t)ch * LCH * 64;
#pragma unroll
    for (int i = 0; i < LCH; ++i) x[i] = p[i * 64];
  };
  for (int t = 0; t < s.ntl; ++t) {
    gf4p tp = s.base + (size_t)t * s.Qq * 64 + lane;
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](const f4 (&x)[LCH], int ch) {
      lf4p bq = bp + ch * LCH * 64;
      const int v = s.Qq - ch * LCH;
      if (v >= LCH) {
#pragma unroll
        for (int i = 0; i < LCH; ++i) {
          const f4 bv = bq[i * 64];
          acc0 = mfma4(x[i][0], bv[0], acc0);
          acc1 = mfma4(x[i][1], bv[1], acc1);
          acc0 = mfma4(x[i][2], bv[2], acc0);
          acc1 = mfma4(x[i][3], bv[3], acc1);
        }
      } else {
#pragma unroll
        for (int i = 0; i < LCH; ++i)
          if (i < v) {
            const f4 bv = bq[i * 64];
            acc0 = mfma4(x[i][0], bv[0], acc0);
            acc1 = mfma4(x[i][1], bv[1], acc1);
            acc0 = mfma4(x[i][2], bv[2], acc0);
            acc1 = mfma4(x[i][3], bv[3], acc1);
          }
      }
    };
    f4 xa[LCH], xb[LCH];
    loadc(xa, tp, 0);
    pre(s.mt0 + t);
    for (int ch = 0;; ch += 2) {
      loadc(xb, tp, ch + 1 < NCH ? ch + 1 : ch);         // (no next chunk: this one again, unused)
      compute(xa, ch);
      if (ch + 1 >= NCH) break;
      loadc(xa, tp, ch + 2 < NCH ? ch + 2 : ch + 1);
      compute(xb, ch + 1);
      if (ch + 2 >= NCH) break;
    }
    epi(s.mt0 + t, acc0 + acc1);
  }
}

// Diagnostic build (-DNJ_GEN_STAMPS): block 0 / thread 0 accumulates s_memtime deltas of the
// phases of an Euler step into g_gen_stamps (read back by tools/ubench/gen_split.py through
// njode_gen_debug_stamps); in the product build no stamp executes.
#ifdef NJ_GEN_STAMPS
__device__ unsigned long long g_gen_stamps[16];
NJ_DEV void gstamp(int slot, unsigned long long& t) {
  const unsigned long long now = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0 && slot >= 0) g_gen_stamps[slot] += now - t;
  t = now;
}
#define GSTAMP(slot, t) gstamp(slot, t)
#else
#define GSTAMP(slot, t)
#endif

// ---- forward of one network on the tile -----------------------------------------------------
// `in` holds the n_in input rows AND the constant-1 row behind them (the input builders below
// write it), complete before the caller's barrier; returns the image that holds the n_out output
// rows.  One barrier per layer.  rec != null: the layer inputs are stored to the evaluation
// record (training calls).  dbase: per-chain dropout hash base of this evaluation (lane's chain
// = lane & 15)
template <class NET>
NJ_DEV lfp net_forward(const GArgs& a, const NET& N, lfp in, lfp other, float* rec, bool drop,
                       uint32_t dbase) {
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
#ifdef NJ_GEN_STAMPS
  unsigned long long ts = __builtin_amdgcn_s_memtime();
#endif
  for (int l = 0; l < N.nl; ++l) {
    const GLayer L = N.l[l];
    const bool hidden = l + 1 < N.nl;
    const FragStream cur = frag_stream(a.frag + L.f_off, L.Qp >> 2, L.MT, L.per);
    if (hidden && tid < 16) other[pix(L.n_out * 16 + tid)] = 1.0f;      // bias unit of the next layer
    if (rec && !ABL(a.dbg, 8)) img_store(rec + (size_t)L.a_row * 16, in, L.n_in);
    GSTAMP(4, ts);
    layer_product(cur, in, [](int) {}, [&](int mt, const f32x4& acc) {
      lfp op = other + (mt << 8) + (c << 2) + g;
      const int u0 = 16 * mt + 4 * g;
      float v[4] = {acc[0], acc[1], acc[2], acc[3]};
      // (the four values of a lane are independent: straight-line code so that they interleave)
      if (hidden && !ABL(a.dbg, 1)) {
        if (L.act == ACT_TANH) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = tanh_acc(v[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
        }
        if (drop) {
#pragma unroll
          for (int r = 0; r < 4; r += 2) {          // (u0 is a multiple of 4)
            const uint32_t w = drop_word(dbase, l, (u0 + r) >> 1);
            v[r] = (w & 0xffffu) >= a.dc.thr16 ? v[r] * a.dc.inv_keep : -0.0f;
            v[r + 1] = (w >> 16) >= a.dc.thr16 ? v[r + 1] * a.dc.inv_keep : -0.0f;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (u0 + r < L.n_out) op[r << 6] = v[r];
    }, a.dbg);
    GSTAMP(5 + (l < 2 ? l : 2), ts);
    if (!ABL(a.dbg, 64)) __syncthreads();
    GSTAMP(8, ts);
    lfp t = in; in = other; other = t;
  }
  return in;
}

// ---- adjoint of one network evaluation ------------------------------------------------------
// `din` holds the delta of the network output (n_out rows; zero for chains that take no part),
// complete before the caller's barrier.  Stores the delta of every layer output to the record
// (for the weight gradients) and leaves the gradient w.r.t. the network INPUT vector in the
// returned image (n_in rows), unless !need_input (then the first layer's transposed product is
// skipped).  The stored activations a layer's epilogue needs (act' and the keep bits) are loaded by
// the lane that uses them, right behind its tile's fragment loads.  One barrier per layer.
template <class NET>
NJ_DEV lfp net_backward(const GArgs& a, const NET& N, lfp din, lfp other, float* rec, bool drop,
                        bool need_input) {
  const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int l_lo = need_input ? 0 : 1;                    // lowest layer whose transposed product runs
  for (int l = N.nl - 1; l >= 0; --l) {
    const GLayer L = N.l[l];
    img_store(rec + (size_t)L.d_row * 16, din, L.n_out);
    if (l < l_lo) break;
    // rows up to the padded k range must be finite (they meet zero fragments): the caller's image
    // here, the image this layer produces below
    if (l == N.nl - 1) {
      for (int e = L.n_out * 16 + tid; e < L.QTp * 64; e += nth) din[pix(e)] = 0.0f;
      __syncthreads();
    }
    if (l - 1 >= l_lo)
      for (int e = L.n_in * 16 + tid; e < N.l[l - 1].QTp * 64; e += nth) other[pix(e)] = 0.0f;
    const float* acts = rec + (size_t)L.a_row * 16;        // this layer's input = act of layer l-1
    const FragStream cur = frag_stream(a.frag + L.ft_off, L.QTp >> 2, L.MTT, L.pert);
    const int pact = l > 0 ? N.l[l - 1].act : -1;
    float av[4] = {0.f, 0.f, 0.f, 0.f};
    layer_product(cur, din, [&](int mt) {
      if (l > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int u = 16 * mt + 4 * g + r;
          av[r] = acts[(u < L.n_in ? u : 0) * 16 + c];
        }
      }
    }, [&](int mt, const f32x4& acc) {
      lfp op = other + (mt << 8) + (c << 2) + g;
      const int u0 = 16 * mt + 4 * g;
      float v[4] = {acc[0], acc[1], acc[2], acc[3]};
      if (l > 0) {
        if (drop) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool dropped = __float_as_uint(av[r]) == 0x80000000u;
            v[r] = dropped ? 0.0f : v[r] * a.dc.inv_keep * dact_rt(pact, av[r] * a.keep);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= dact_rt(pact, av[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (u0 + r < L.n_in) op[r << 6] = v[r];
    });
    __syncthreads();
    lfp t = din; din = other; other = t;
  }
  return din;
}

// ---- small tile helpers ---------------------------------------------------------------------
// encoder input [tanh(x) ; mask] into `in` (models.py:261-276: ffnn(cat(tanh(x), mask)))
NJ_DEV void enc_input(const GArgs& a, lfp in, lfp x, lfp mask) {
  for (int e = threadIdx.x; e < a.D * 16; e += blockDim.x) {
    in[pix(e)] = tanh_acc(x[e]);
    if (a.masked) in[pix(a.D * 16 + e)] = mask[e];
  }
  if (threadIdx.x < 16) in[pix((a.masked ? 2 * a.D : a.D) * 16 + threadIdx.x)] = 1.0f;   // bias unit
}
// identity path of the encoder added to its output: out[j] += f(x)   (raw x, models.py:240-259)
NJ_DEV void enc_residual(const GArgs& a, lfp out, lfp x) {
  if (a.enc_case == 0) return;
  for (int e = threadIdx.x; e < a.H * 16; e += blockDim.x) {
    const int j = e >> 4, c = e & 15;
    float s;
    if (a.enc_case == 1) s = x[(j % a.D) * 16 + c];
    else {
      s = 0.0f;
      for (int q = 0; q < a.enc_mult; ++q) s += x[(q * a.H + j) * 16 + c];
      s *= 1.0f / a.enc_mult;
    }
    out[e] += s;
  }
}
NJ_DEV void dec_input(const GArgs& a, lfp in, lfp h) {
  for (int e = threadIdx.x; e < a.H * 16; e += blockDim.x) in[pix(e)] = tanh_acc(h[e]);
  if (threadIdx.x < 16) in[pix(a.H * 16 + threadIdx.x)] = 1.0f;                            // bias unit
}
NJ_DEV void dec_residual(const GArgs& a, lfp out, lfp h) {
  if (a.dec_case == 0) return;
  for (int e = threadIdx.x; e < a.DO * 16; e += blockDim.x) {
    const int j = e >> 4, c = e & 15;
    float s;
    if (a.dec_case == 1) s = h[(j % a.H) * 16 + c];
    else {
      s = 0.0f;
      for (int q = 0; q < a.dec_mult; ++q) s += h[(q * a.DO + j) * 16 + c];
      s *= 1.0f / a.dec_mult;
    }
    out[e] += s;
  }
}

// paper loss of the tile's observed chains (models.py:71-126) and its gradients; one thread per
// chain.  has[c]: chain c has an observation.  dy / dybj may be null (forward).
NJ_DEV void loss_tile(const GArgs& a, lfp x, lfp mask, lfp y, lfp ybj, const int* rows, lfp scale,
                      lfp loss_acc, lfp dy, lfp dybj) {
  const int c = threadIdx.x;
  if (c < 16) {
    const bool has = rows[c] >= 0;
    float sa = 0.0f, sb = 0.0f;
    for (int q = 0; q < a.D; ++q) {
      const float m = a.masked ? mask[q * 16 + c] : 1.0f;
      const float e = x[q * 16 + c] - y[q * 16 + c];
      const float f = a.loss_easy ? (ybj[q * 16 + c] - x[q * 16 + c]) : (ybj[q * 16 + c] - y[q * 16 + c]);
      sa = fmaf(m * e, e, sa);
      sb = fmaf(m * f, f, sb);
    }
    const float na = sqrtf(sa + 1e-10f), nb = sqrtf(sb + 1e-10f);
    const float w = a.weight;
    const float ca = a.loss_easy ? w : 2.0f * w, cb = a.loss_easy ? (1.0f - w) : 2.0f * (1.0f - w);
    const float s = ca * na + cb * nb;
    const float sc = has ? scale[c] : 0.0f;
    if (loss_acc) loss_acc[c] += s * s * sc;
    if (dy) {
      const float gq = 2.0f * s * sc;
      const float ga = gq * ca / na, gb = gq * cb / nb;
      for (int q = 0; q < a.D; ++q) {
        const float m = a.masked ? mask[q * 16 + c] : 1.0f;
        const float e = x[q * 16 + c] - y[q * 16 + c];
        if (a.loss_easy) {
          const float f = ybj[q * 16 + c] - x[q * 16 + c];
          dy[q * 16 + c] = -ga * m * e;
          dybj[q * 16 + c] = gb * m * f;
        } else {
          const float f = ybj[q * 16 + c] - y[q * 16 + c];
          dy[q * 16 + c] = -ga * m * e - gb * m * f;
          dybj[q * 16 + c] = gb * m * f;
        }
      }
    }
  }
}

// ODE input vector of the tile (models.py:188-199): [tanh(last_X), tanh(h), tau, t - tau (, t)]
NJ_DEV void ode_input(const GArgs& a, lfp in, lfp tx, lfp h, lfp tau, float t) {
  const int n = (a.D + a.H) * 16;
  for (int e = threadIdx.x; e < n; e += blockDim.x)
    in[pix(e)] = e < a.D * 16 ? tx[e] : tanh_acc(h[e - a.D * 16]);
  if (threadIdx.x < 16) {
    const int c = threadIdx.x;
    const float ta = tau[c], td = t - ta;
    in[pix((a.D + a.H) * 16 + c)] = ta;
    in[pix((a.D + a.H + 1) * 16 + c)] = td;
    if (a.curt) in[pix((a.D + a.H + 2) * 16 + c)] = ta + td;
    in[pix(a.IN0 * 16 + c)] = 1.0f;                                                        // bias unit
  }
}

// ---- GRU jump of the tile ---------------------------------------------------------------------
NJ_DEV float sigmoid_acc(float x) { return fmaf(0.5f, tanh_acc(0.5f * x), 0.5f); }
// S.hn <- GRUCell(tanh(S.xr), tanh(S.h)) for every chain (the caller commits the observed ones).
// rec != null: the layer input [tanh x; tanh h] goes to the record's input rows (net_forward) and
// the four gate pre-activations to its DELTA rows, where the sweep reads them before it overwrites
// them with the deltas.
NJ_DEV void gru_jump_fwd(const GArgs& a, GLds& S, float* rec) {
  const int tid = threadIdx.x, nth = blockDim.x, D = a.D, H = a.H;
  for (int e = tid; e < D * 16; e += nth) S.img0[pix(e)] = tanh_acc(S.xr[e]);
  for (int e = tid; e < H * 16; e += nth) S.img0[pix(D * 16 + e)] = tanh_acc(S.h[e]);
  if (tid < 16) S.img0[pix((D + H) * 16 + tid)] = 1.0f;                                      // bias unit
  __syncthreads();
  lfp out = net_forward(a, a.gru, S.img0, S.img1, rec, false, 0u);
  if (rec) img_store(rec + (size_t)a.gru.l[0].d_row * 16, out, 4 * H);
  for (int e = tid; e < H * 16; e += nth) {
    const float r = sigmoid_acc(out[pix(e)]), z = sigmoid_acc(out[pix(H * 16 + e)]);
    const float n = tanh_acc(out[pix(2 * H * 16 + e)] + r * out[pix(3 * H * 16 + e)]);
    const float th = S.img0[pix(D * 16 + e)];
    S.hn[e] = fmaf(z, th - n, n);                                                             // (1 - z) n + z th
  }
  __syncthreads();
}
// adjoint of the jump: lam (= S.hn: adjoint of the new state, zero for chains without an observation)
// -> S.hn = adjoint of the state before the jump through the cell; deltas into the record.
NJ_DEV void gru_jump_bwd(const GArgs& a, GLds& S, float* rec) {
  const int tid = threadIdx.x, nth = blockDim.x, D = a.D, H = a.H;
  const float* pre = rec + (size_t)a.gru.l[0].d_row * 16;       // the forward's gate pre-activations
  const float* gin = rec + (size_t)a.gru.l[0].a_row * 16;       // [tanh x; tanh h]
  for (int e = tid; e < H * 16; e += nth) {
    const float lam = S.hn[e];
    const float gh = pre[3 * H * 16 + e], th = gin[D * 16 + e];
    const float r = sigmoid_acc(pre[e]), z = sigmoid_acc(pre[H * 16 + e]);
    const float n = tanh_acc(pre[2 * H * 16 + e] + r * gh);
    const float dnp = lam * (1.0f - z) * (1.0f - n * n);        // d / d (pre-activation of n)
    S.img0[pix(e)] = dnp * gh * r * (1.0f - r);
    S.img0[pix(H * 16 + e)] = lam * (th - n) * z * (1.0f - z);
    S.img0[pix(2 * H * 16 + e)] = dnp;
    S.img0[pix(3 * H * 16 + e)] = dnp * r;
    S.hn[e] = lam * z * (1.0f - th * th);                       // the direct path h' = ... + z tanh(h)
  }
  __syncthreads();
  lfp din = net_backward(a, a.gru, S.img0, S.img1, rec, false, true);
  for (int e = tid; e < H * 16; e += nth) {
    const float th = gin[D * 16 + e];
    S.hn[e] = fmaf(din[pix(D * 16 + e)], 1.0f - th * th, S.hn[e]);
  }
  __syncthreads();
}

// =============================================================================================
// forward: one workgroup per tile of 16 paths (models.py:379-518)
// =============================================================================================
__global__ void __launch_bounds__(1024) k_gen_fwd(GArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  GLds S;
  S.carve((lfp)smem, a);
  const int tid = threadIdx.x, nth = blockDim.x;
  // (round 4: a tile holds a.PT <= 16 paths, the other chains idle -- see gen_paths_per_tile)
  const int tile = blockIdx.x, b0 = tile * a.PT;
  const int nv = a.B - b0 < a.PT ? a.B - b0 : a.PT;           // paths of this tile
  {
    const int n = gen_lds_floats(a.img_rows, a.D, a.H, a.DO);
    for (int e = tid; e < n; e += nth) ((lfp)smem)[e] = 0.0f;
  }
  __syncthreads();
  lfp tau = S.misc, lossacc = S.misc + 16, scale = S.misc + 32;
  const int cch = tid & 15;                                  // chain of this lane in epilogues
  const int bch = cch < nv ? b0 + cch : a.B - 1;
  const unsigned long long gidc = a.gid0 + (unsigned long long)bch;
  if (tid < 16) {
    const int b = b0 + tid;
    scale[tid] = (tid < nv && a.want_loss) ? a.inv_batch / (float)a.n_obs_ot[b] : 0.0f;
  }
  // ---- h = encoder(start_X) with a zero mask (models.py:411-414)
  for (int e = tid; e < a.D * 16; e += nth) {
    const int q = e >> 4, c = e & 15, b = b0 + c;
    const float v = c < nv ? a.start_X[(size_t)b * a.D + q] : 0.0f;
    S.xr[e] = v;
    S.tx[e] = tanh_acc(v);
    S.mk[e] = 0.0f;
  }
  __syncthreads();
  enc_input(a, S.img0, S.xr, S.mk);
  __syncthreads();
  {
    float* rec = a.save ? a.rec_enc + ((size_t)a.n_times * a.T + tile) * a.enc.rec_rows * 16 : nullptr;
    lfp out = net_forward(a, a.enc, S.img0, S.img1, rec, a.drop != 0,
                          drop_base(a.dc, gidc, G_TKEY_START, G_NET_ENC));
    img_get(S.h, out, a.H);
    __syncthreads();
    enc_residual(a, S.h, S.xr);
    __syncthreads();
  }
  int prow = 0;
  auto emit_row = [&](uint32_t tkey) {   // path output: readout of the current state
    if (!a.want_path) return;
    dec_input(a, S.img0, S.h);
    __syncthreads();
    lfp out = net_forward(a, a.dec, S.img0, S.img1, nullptr, a.drop != 0,
                          drop_base(a.dc, gidc, tkey, G_NET_DEC_ROW));
    img_get(S.y, out, a.DO);
    __syncthreads();
    dec_residual(a, S.y, S.h);
    __syncthreads();
  };
  auto write_row = [&]() {
    if (!a.want_path) return;
    float* ph = a.path_h + ((size_t)prow * a.B + b0) * a.H;
    for (int e = tid; e < nv * a.H; e += nth) ph[e] = S.h[(e % a.H) * 16 + e / a.H];
    float* py = a.path_y + ((size_t)prow * a.B + b0) * a.DO;
    for (int e = tid; e < nv * a.DO; e += nth) py[e] = S.y[(e % a.DO) * 16 + e / a.DO];
    ++prow;
  };
  emit_row(G_TKEY_START - 1);
  write_row();

  // the row of every chain at the NEXT jump time is fetched one jump ahead (jump times are visited
  // in increasing order): its latency hides behind the evaluations in between
  int row_pf = -1;
  auto prefetch_rows = [&](int i) {
    if (tid < 16)
      row_pf = (i < a.n_times && tid < nv && a.n_obs > 0) ? a.dense[(size_t)i * a.B + b0 + tid] : -1;
  };
  prefetch_rows(0);
  constexpr int XR = 4;                       // observation values held in registers per thread
  for (int k = 0;; ++k) {
    // ---- jumps that happen before Euler step k
    for (int i = a.jlo[k]; i < a.jlo[k + 1]; ++i) {
      if (tid < 16) S.rows[tid] = row_pf;
      prefetch_rows(i + 1);
      __syncthreads();
      bool any = false;
#pragma unroll
      for (int c = 0; c < 16; ++c) any |= S.rows[c] >= 0;
      if (any) {
        const size_t jrec = (size_t)i * a.T + tile;
        // observation and mask: loads issued now, consumed after the first readout
        float xv[XR] = {0.f, 0.f, 0.f, 0.f}, mv[XR] = {0.f, 0.f, 0.f, 0.f};
        const bool x_regs = a.D * 16 <= XR * nth;
        if (x_regs) {
#pragma unroll
          for (int q = 0; q < XR; ++q) {
            const int e = tid + q * nth;
            const int ee = e < a.D * 16 ? e : 0;
            const int r = S.rows[ee & 15];
            const bool on = e < a.D * 16 && r >= 0;
            const size_t src = (size_t)(on ? r : 0) * a.D + (ee >> 4);
            xv[q] = on ? a.X[src] : 0.0f;
            mv[q] = a.masked ? (on ? a.M[src] : 0.0f) : 1.0f;
          }
        }
        // y_bj = readout(h)
        dec_input(a, S.img0, S.h);
        __syncthreads();
        {
          float* rec = a.save ? a.rec_dec + (jrec * 2 + 0) * a.dec.rec_rows * 16 : nullptr;
          lfp out = net_forward(a, a.dec, S.img0, S.img1, rec, a.drop != 0,
                                drop_base(a.dc, gidc, (uint32_t)k, G_NET_DEC_BJ));
          img_get(S.ybj, out, a.DO);
          __syncthreads();
          dec_residual(a, S.ybj, S.h);
        }
        // observation, mask, encoder input (self-imputation in masked mode, models.py:463-467)
        if (x_regs) {
#pragma unroll
          for (int q = 0; q < XR; ++q) {
            const int e = tid + q * nth;
            if (e < a.D * 16) {
              S.xr[e] = xv[q];
              S.mk[e] = mv[q];
            }
          }
        } else {
          for (int e = tid; e < a.D * 16; e += nth) {
            const int q = e >> 4, c = e & 15, r = S.rows[c];
            const float xvv = r >= 0 ? a.X[(size_t)r * a.D + q] : 0.0f;
            const float mvv = a.masked ? (r >= 0 ? a.M[(size_t)r * a.D + q] : 0.0f) : 1.0f;
            S.xr[e] = xvv;
            S.mk[e] = mvv;
          }
        }
        __syncthreads();
        if (a.rnn) {
          // h_new = GRUCell(tanh(X_obs), tanh(h))  (models.py:202-217; unmasked models only)
          float* rec = a.save ? a.rec_gru + jrec * a.gru.rec_rows * 16 : nullptr;
          gru_jump_fwd(a, S, rec);
        } else {
          for (int e = tid; e < a.D * 16; e += nth)
            S.xin[e] = a.masked ? S.xr[e] * S.mk[e] + (1.0f - S.mk[e]) * S.ybj[e] : S.xr[e];
          __syncthreads();
          enc_input(a, S.img0, S.xin, S.mk);
          __syncthreads();
          float* rec = a.save ? a.rec_enc + jrec * a.enc.rec_rows * 16 : nullptr;
          lfp out = net_forward(a, a.enc, S.img0, S.img1, rec, a.drop != 0,
                                drop_base(a.dc, gidc, (uint32_t)k, G_NET_ENC));
          img_get(S.hn, out, a.H);
          __syncthreads();
          enc_residual(a, S.hn, S.xin);
          __syncthreads();
        }
        // y = readout(h_new); chains without an observation keep their state: evaluate the
        // readout on the state they will have after the commit
        for (int e = tid; e < a.H * 16; e += nth)
          if (S.rows[e & 15] < 0) S.hn[e] = S.h[e];
        __syncthreads();
        dec_input(a, S.img0, S.hn);
        __syncthreads();
        {
          float* rec = a.save ? a.rec_dec + (jrec * 2 + 1) * a.dec.rec_rows * 16 : nullptr;
          lfp out = net_forward(a, a.dec, S.img0, S.img1, rec, a.drop != 0,
                                drop_base(a.dc, gidc, (uint32_t)k, G_NET_DEC));
          // (path output: unobserved chains keep the y of their last row, as the reference's
          // whole-batch readout gives them in eval mode)
          for (int e = tid; e < a.DO * 16; e += nth)
            if (S.rows[e & 15] >= 0 || !a.want_path) S.y[e] = out[pix(e)];
          __syncthreads();
          for (int e = tid; e < a.DO * 16; e += nth) {
            if (S.rows[e & 15] >= 0 || !a.want_path) {
              const int j = e >> 4, c = e & 15;
              float s = 0.0f;
              if (a.dec_case == 1) s = S.hn[(j % a.H) * 16 + c];
              else if (a.dec_case == 2) {
                for (int q = 0; q < a.dec_mult; ++q) s += S.hn[(q * a.DO + j) * 16 + c];
                s *= 1.0f / a.dec_mult;
              }
              S.y[e] += s;
            }
          }
          __syncthreads();
        }
        if (a.want_loss) loss_tile(a, S.xr, S.mk, S.y, S.ybj, S.rows, scale, lossacc, nullptr, nullptr);
        if (a.save) {
          float* yb = a.ybuf + jrec * 2 * a.DO * 16;
          for (int e = tid; e < a.DO * 16; e += nth) {
            yb[e] = S.ybj[e];
            yb[a.DO * 16 + e] = S.y[e];
          }
          if (tid == 0) a.flags[jrec] = 1;
        }
        // commit (models.py:468-489): h <- h_new, last_X <- X_obs (masked: Y), tau <- obs time
        for (int e = tid; e < a.H * 16; e += nth)
          if (S.rows[e & 15] >= 0) S.h[e] = S.hn[e];
        for (int e = tid; e < a.D * 16; e += nth)
          if (S.rows[e & 15] >= 0) S.tx[e] = tanh_acc(a.masked ? S.y[e] : S.xr[e]);
        if (tid < 16 && S.rows[tid] >= 0) tau[tid] = a.time_f32[i];
        __syncthreads();
      }
      write_row();
      __syncthreads();
    }
    if (k >= a.K) break;
    // ---- Euler step k (models.py:369-377)
    {
#ifdef NJ_GEN_STAMPS
      unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
      const float dt = a.step_dt[k], t = a.step_t[k];
      ode_input(a, S.img0, S.tx, S.h, tau, t);
      __syncthreads();
      GSTAMP(0, t0);
      float* rec = a.save ? a.rec_ode + ((size_t)k * a.T + tile) * a.ode.rec_rows * 16 : nullptr;
      lfp out = net_forward(a, a.ode, S.img0, S.img1, rec, a.drop != 0,
                            drop_base(a.dc, gidc, (uint32_t)k, G_NET_ODE));
      GSTAMP(1, t0);
      for (int e = tid; e < a.H * 16; e += nth) S.h[e] = fmaf(dt, out[pix(e)], S.h[e]);
      __syncthreads();
      GSTAMP(2, t0);
      emit_row(0x80000000u + (uint32_t)k);
      write_row();
    }
  }
  if (a.hT) {
    float* ph = a.hT + (size_t)b0 * a.H;
    for (int e = tid; e < nv * a.H; e += nth) ph[e] = S.h[(e % a.H) * 16 + e / a.H];
  }
  if (a.want_loss && tid < nv) a.loss_terms[b0 + tid] = lossacc[tid];
}

// =============================================================================================
// adjoint sweep: events of the tile in reverse (same mathematics as njode_lockstep_bwd.h)
// =============================================================================================
__global__ void __launch_bounds__(1024) k_gen_bwd(GArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  GLds S;
  S.carve((lfp)smem, a);
  const int tid = threadIdx.x, nth = blockDim.x;
  const int tile = blockIdx.x, b0 = tile * a.PT;
  const int nv = a.B - b0 < a.PT ? a.B - b0 : a.PT;           // paths of this tile
  {
    const int n = gen_lds_floats(a.img_rows, a.D, a.H, a.DO);
    for (int e = tid; e < n; e += nth) ((lfp)smem)[e] = 0.0f;
  }
  __syncthreads();
  lfp scale = S.misc + 32;
  if (tid < 16) {
    const int b = b0 + tid;
    scale[tid] = tid < nv ? a.inv_batch / (float)a.n_obs_ot[b] : 0.0f;
  }
  // adjoints: lam_h = S.h [H], lam_x = S.tx [D] (w.r.t. last_X, masked models)
  lfp lam_h = S.h, lam_x = S.tx, dy = S.y, dybj = S.ybj, lam_hn = S.hn;
  const bool drop = a.drop != 0;
  if (a.g_hT) {   // the adjoint of the final state starts from the upstream gradient of hT
    for (int e = tid; e < a.H * 16; e += nth) {
      const int j = e >> 4, c = e & 15;
      lam_h[e] = c < nv ? a.g_hT[(size_t)(b0 + c) * a.H + j] : 0.0f;
    }
  }
  __syncthreads();
  for (int k = a.K; k >= 0; --k) {
    if (k < a.K) {
      // ---- reverse Euler step k: h' = h + dt f(in0)
      float* rec = a.rec_ode + ((size_t)k * a.T + tile) * a.ode.rec_rows * 16;
      const float dt = a.step_dt[k];
      for (int e = tid; e < a.H * 16; e += nth) S.img0[pix(e)] = dt * lam_h[e];
      __syncthreads();
      lfp din = net_backward(a, a.ode, S.img0, S.img1, rec, drop, true);
      const float* in0 = rec + (size_t)a.ode.l[0].a_row * 16;
      for (int e = tid; e < a.H * 16; e += nth) {
        const float th = in0[a.D * 16 + e];
        lam_h[e] = fmaf(din[pix(a.D * 16 + e)], 1.0f - th * th, lam_h[e]);
      }
      if (a.masked) {
        for (int e = tid; e < a.D * 16; e += nth) {
          const float t = in0[e];
          lam_x[e] = fmaf(din[pix(e)], 1.0f - t * t, lam_x[e]);
        }
      }
      __syncthreads();
    }
    // ---- reverse the jumps the forward applied right before step k
    for (int i = a.jlo[k + 1] - 1; i >= a.jlo[k]; --i) {
      const size_t jrec = (size_t)i * a.T + tile;
      if (!a.flags[jrec]) continue;                           // workgroup-uniform
      if (tid < 16) S.rows[tid] = tid < nv ? a.dense[(size_t)i * a.B + b0 + tid] : -1;
      // observation, mask, stored readouts
      {
        const float* yb = a.ybuf + jrec * 2 * a.DO * 16;
        __syncthreads();
        for (int e = tid; e < a.D * 16; e += nth) {
          const int q = e >> 4, c = e & 15, r = S.rows[c];
          S.xr[e] = r >= 0 ? a.X[(size_t)r * a.D + q] : 0.0f;
          S.mk[e] = a.masked ? (r >= 0 ? a.M[(size_t)r * a.D + q] : 0.0f) : 1.0f;
          S.xin[e] = yb[e];                 // y_bj
          S.img1[e] = yb[a.DO * 16 + e];    // y
        }
        __syncthreads();
        loss_tile(a, S.xr, S.mk, S.img1, S.xin, S.rows, scale, nullptr, dy, dybj);
        __syncthreads();
      }
      // unobserved chains: no loss, and their state passed through the jump unchanged
      for (int e = tid; e < a.DO * 16; e += nth) {
        const bool has = S.rows[e & 15] >= 0;
        float v = has ? dy[e] : 0.0f;
        if (has && a.masked) v += lam_x[e];     // last_X <- Y: the next segment's input gradient
        dy[e] = v;
        if (!has) dybj[e] = 0.0f;
      }
      __syncthreads();
      // y = readout(h_new)
      float* rec_y = a.rec_dec + (jrec * 2 + 1) * a.dec.rec_rows * 16;
      img_put(S.img0, dy, a.DO);
      __syncthreads();
      {
        lfp din = net_backward(a, a.dec, S.img0, S.img1, rec_y, drop, true);
        const float* th = rec_y + (size_t)a.dec.l[0].a_row * 16;
        for (int e = tid; e < a.H * 16; e += nth) {
          const int j = e >> 4, c = e & 15;
          if (S.rows[c] >= 0) {
            const float t = th[e];
            float v = din[pix(e)] * (1.0f - t * t);
            if (a.dec_case == 1) {
              for (int q = j; q < a.DO; q += a.H) v += dy[q * 16 + c];
            } else if (a.dec_case == 2) {
              v += dy[(j % a.DO) * 16 + c] * (1.0f / a.dec_mult);
            }
            lam_hn[e] = lam_h[e] + v;
          } else {
            lam_hn[e] = 0.0f;
          }
        }
        __syncthreads();
      }
      // h_new = GRUCell(tanh x, tanh h_pre): a second path into the state before the jump (left in
      // lam_hn for the commit below)
      if (a.rnn) gru_jump_bwd(a, S, a.rec_gru + jrec * a.gru.rec_rows * 16);
      // h_new = encoder(x_in, M)
      float* rec_e = a.rec_enc + jrec * a.enc.rec_rows * 16;
      if (!a.rnn) img_put(S.img0, lam_hn, a.H);
      __syncthreads();
      if (!a.rnn) {
        lfp din = net_backward(a, a.enc, S.img0, S.img1, rec_e, drop, a.masked != 0);
        if (a.masked) {
          const float* ein = rec_e + (size_t)a.enc.l[0].a_row * 16;
          for (int e = tid; e < a.D * 16; e += nth) {
            const int q = e >> 4, c = e & 15;
            if (S.rows[c] >= 0) {
              const float t = ein[e];
              float v = din[pix(e)] * (1.0f - t * t);
              if (a.enc_case == 1) {
                for (int j = q; j < a.H; j += a.D) v += lam_hn[j * 16 + c];
              } else if (a.enc_case == 2) {
                v += lam_hn[(q % a.H) * 16 + c] * (1.0f / a.enc_mult);
              }
              dybj[e] += v * (1.0f - S.mk[e]);
            }
          }
        }
        __syncthreads();
      }
      // y_bj = readout(h_pre): the only path from the state before the jump
      float* rec_b = a.rec_dec + (jrec * 2 + 0) * a.dec.rec_rows * 16;
      img_put(S.img0, dybj, a.DO);
      __syncthreads();
      {
        lfp din = net_backward(a, a.dec, S.img0, S.img1, rec_b, drop, true);
        const float* th = rec_b + (size_t)a.dec.l[0].a_row * 16;
        for (int e = tid; e < a.H * 16; e += nth) {
          const int j = e >> 4, c = e & 15;
          if (S.rows[c] >= 0) {
            const float t = th[e];
            float v = din[pix(e)] * (1.0f - t * t);
            if (a.dec_case == 1) {
              for (int q = j; q < a.DO; q += a.H) v += dybj[q * 16 + c];
            } else if (a.dec_case == 2) {
              v += dybj[(j % a.DO) * 16 + c] * (1.0f / a.dec_mult);
            }
            lam_h[e] = a.rnn ? v + lam_hn[e] : v;
          }
        }
        for (int e = tid; e < a.D * 16; e += nth)
          if (S.rows[e & 15] >= 0) lam_x[e] = 0.0f;
        __syncthreads();
      }
    }
  }
  // ---- start state: h0 = encoder(start_X): its deltas for the weight gradients
  {
    float* rec = a.rec_enc + ((size_t)a.n_times * a.T + tile) * a.enc.rec_rows * 16;
    img_put(S.img0, lam_h, a.H);
    __syncthreads();
    (void)net_backward(a, a.enc, S.img0, S.img1, rec, drop, false);
  }
}

// =============================================================================================
// weight gradients: dW[o][i] = sum over records and chains of delta[o][chain] * input[i][chain]
// =============================================================================================
struct GDw {
  const float* rec;      // records of this network
  long long n_rec;       // number of records
  const int* n_rec_dev;  // non-null: the number of records lives on the device (segment plan)
  int rec_floats;        // floats per record
  const int* flags;      // per flag group 0 / 1, or null (all active)
  int flag_div;          // record r belongs to flag group r / flag_div
  long long n_flagged;   // records [0, n_flagged) are flagged, the rest always active
  int a_row, d_row, n_in, n_out;
  int w_off, b_off;      // destination offsets in a slab row
  int kind;              // GLayer.kind (1: the GRU cell's combined layer, gru_dw_store)
  int P;                 // slab row length
  int tiles_m, tiles_n;  // 16-unit tiles over outputs / inputs (+ bias column)
  float* slab;           // [gridDim.y][P]
};
constexpr int DW_TM = 4, DW_TN = 4;
__global__ void __launch_bounds__(256) k_gen_dw(GDw d) {
  __shared__ __attribute__((aligned(16))) float red[3 * DW_TM * DW_TN * 4 * 64];
  const int lane = threadIdx.x & 63, wv = wave_id(), g = lane >> 4, c = lane & 15;
  const int bn = (d.tiles_n + DW_TN - 1) / DW_TN;
  const int tm0 = (blockIdx.x / bn) * DW_TM, tn0 = (blockIdx.x % bn) * DW_TN;
  f32x4 G[DW_TM][DW_TN];
#pragma unroll
  for (int i = 0; i < DW_TM; ++i)
#pragma unroll
    for (int j = 0; j < DW_TN; ++j) G[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long stride = (long long)gridDim.y * 4;
  const int nm = d.tiles_m - tm0 < DW_TM ? d.tiles_m - tm0 : DW_TM;     // tiles of this block (uniform)
  const int nn = d.tiles_n - tn0 < DW_TN ? d.tiles_n - tn0 : DW_TN;
  // operand rows of this lane: clamp to a valid row, zero the value afterwards
  int arow[DW_TM], brow[DW_TN];
  bool aok[DW_TM], bok[DW_TN], bone[DW_TN];
#pragma unroll
  for (int i = 0; i < DW_TM; ++i) {
    const int u = 16 * (tm0 + i) + c;
    aok[i] = u < d.n_out;
    arow[i] = d.d_row + (aok[i] ? u : 0);
  }
#pragma unroll
  for (int j = 0; j < DW_TN; ++j) {
    const int u = 16 * (tn0 + j) + c;
    bok[j] = u < d.n_in;
    bone[j] = u == d.n_in;            // bias column: input = 1
    brow[j] = d.a_row + (bok[j] ? u : 0);
  }
  const long long n_rec = d.n_rec_dev ? (long long)d.n_rec_dev[0] : d.n_rec;
  // records of this wave: r0, r0 + stride, ...; the operands of the next record are in flight
  // while the 64 MFMAs of the current one issue (two register sets, no copies)
  auto next_active = [&](long long r) {
    while (r < n_rec && d.flags && r < d.n_flagged && !d.flags[r / d.flag_div]) r += stride;   // wave-uniform
    return r;
  };
  auto load = [&](long long r, f4 (&af)[DW_TM], f4 (&bf)[DW_TN]) {
    const float* rec = d.rec + (size_t)r * d.rec_floats;
#pragma unroll
    for (int i = 0; i < DW_TM; ++i) af[i] = *(const f4*)(rec + (size_t)arow[i] * 16 + 4 * g);
#pragma unroll
    for (int j = 0; j < DW_TN; ++j) bf[j] = *(const f4*)(rec + (size_t)brow[j] * 16 + 4 * g);
  };
  auto mm = [&](f4 (&af)[DW_TM], f4 (&bf)[DW_TN]) {
#pragma unroll
    for (int i = 0; i < DW_TM; ++i)
      if (!aok[i]) af[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < DW_TN; ++j)
      if (!bok[j]) bf[j] = bone[j] ? f4{1.f, 1.f, 1.f, 1.f} : f4{0.f, 0.f, 0.f, 0.f};
    // (tiles beyond the layer's edge are skipped: a narrow first / last layer has one tile column / row)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < DW_TM; ++i)
        if (i < nm) {
#pragma unroll
          for (int j = 0; j < DW_TN; ++j)
            if (j < nn) G[i][j] = mfma4(af[i][s], bf[j][s], G[i][j]);
        }
  };
  long long r = next_active((long long)blockIdx.y * 4 + wv);
  if (r < n_rec) {
    f4 a0[DW_TM], b0[DW_TN], a1[DW_TM], b1[DW_TN];
    load(r, a0, b0);
    for (;;) {
      const long long r1 = next_active(r + stride);
      const bool h1 = r1 < n_rec;
      load(h1 ? r1 : r, a1, b1);          // (no next record: the same one again, not used)
      mm(a0, b0);
      if (!h1) break;
      const long long r2 = next_active(r1 + stride);
      const bool h2 = r2 < n_rec;
      load(h2 ? r2 : r1, a0, b0);
      mm(a1, b1);
      if (!h2) break;
      r = r2;
    }
  }
  // the four waves of the block add their tiles in fixed order -> one slab row per blockIdx.y
  f32x4 __attribute__((address_space(3)))* rd = (f32x4 __attribute__((address_space(3)))*)(lfp)red;
  if (wv > 0) {
#pragma unroll
    for (int i = 0; i < DW_TM; ++i)
#pragma unroll
      for (int j = 0; j < DW_TN; ++j) rd[((wv - 1) * DW_TM * DW_TN + i * DW_TN + j) * 64 + lane] = G[i][j];
  }
  __syncthreads();
  if (wv != 0) return;
  float* row = d.slab + (size_t)blockIdx.y * d.P;
#pragma unroll
  for (int i = 0; i < DW_TM; ++i)
#pragma unroll
    for (int j = 0; j < DW_TN; ++j) {
      f32x4 t = G[i][j];
      t += rd[(0 * DW_TM * DW_TN + i * DW_TN + j) * 64 + lane];
      t += rd[(1 * DW_TM * DW_TN + i * DW_TN + j) * 64 + lane];
      t += rd[(2 * DW_TM * DW_TN + i * DW_TN + j) * 64 + lane];
      const int ui = 16 * (tn0 + j) + c;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int uo = 16 * (tm0 + i) + 4 * g + rr;
        if (uo < d.n_out) {
          if (d.kind == 1) { if (ui <= d.n_in) gru_dw_store(row, d.w_off, d.n_in, d.n_out, uo, ui, t[rr]); }
          else if (ui < d.n_in) row[d.w_off + (size_t)uo * d.n_in + ui] = t[rr];
          else if (ui == d.n_in) row[d.b_off + uo] = t[rr];
        }
      }
    }
}

// ---- fragment tables ------------------------------------------------------------------------
struct GPack { int n_layers; GLayer l[3 * GEN_MAXL + 1]; int total; };
__global__ void k_gen_pack(const float* __restrict__ P, float* __restrict__ frag, GPack p) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.total) return;
  for (int q = 0; q < p.n_layers; ++q) {
    const GLayer L = p.l[q];
    const int nf = L.MT * L.Qp * 64, nt = L.MTT * L.QTp * 64;
    if (idx >= L.f_off && idx < L.f_off + nf) {
      // [tile][quad][lane][k-step of the quad]
      const int e = idx - L.f_off, qq = e & 3, lane = (e >> 2) & 63, f = e >> 8, Qq = L.Qp >> 2;
      const int mt = f / Qq, k = 4 * (f % Qq) + qq;
      const int o = 16 * mt + (lane & 15), i = 4 * k + (lane >> 4);
      float v = 0.0f;
      if (o < L.n_out && i <= L.n_in)
        v = L.kind == 1 ? gru_w(P, L, o, i)
                        : (i < L.n_in ? P[L.w_off + (size_t)o * L.n_in + i] : P[L.b_off + o]);
      frag[idx] = v;
      return;
    }
    if (idx >= L.ft_off && idx < L.ft_off + nt) {
      const int e = idx - L.ft_off, qq = e & 3, lane = (e >> 2) & 63, f = e >> 8, Qq = L.QTp >> 2;
      const int mt = f / Qq, k = 4 * (f % Qq) + qq;
      const int i = 16 * mt + (lane & 15), o = 4 * k + (lane >> 4);
      frag[idx] = (i < L.n_in && o < L.n_out)
                      ? (L.kind == 1 ? gru_w(P, L, o, i) : P[L.w_off + (size_t)o * L.n_in + i]) : 0.0f;
      return;
    }
  }
}

// ---- plan: time index of every row, dense [time][path] -> row, first jump of every step ------
__global__ void k_gen_rows(const int* __restrict__ time_ptr, int n_times, int n_obs,
                           const int* __restrict__ obs_idx, int B, int* __restrict__ dense,
                           int* __restrict__ bad) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_obs) return;
  int lo = 0, hi = n_times;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (time_ptr[mid] <= r) lo = mid; else hi = mid;
  }
  const int b = obs_idx[r];
  if (b < 0 || b >= B) { atomicOr(bad, 1); return; }
  dense[(size_t)lo * B + b] = r;
}
__global__ void k_gen_jlo(const int* __restrict__ k_jump, int n_times, int K, int* __restrict__ jlo) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > K + 1) return;
  int lo = 0, hi = n_times;         // first i with k_jump[i] >= k
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (k_jump[mid] < k) lo = mid + 1; else hi = mid;
  }
  jlo[k] = lo;
}

// grad[p] = grad_loss * sum_s slab[s][p]  (fixed order)
__global__ void k_gen_reduce(const float* __restrict__ slab, int S, int P,
                             const float* __restrict__ grad_loss, float* __restrict__ grad) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float acc = 0.0f;
  for (int s = 0; s < S; ++s) acc += slab[(size_t)s * P + p];
  grad[p] = acc * grad_loss[0];
}
__global__ void __launch_bounds__(1024) k_gen_sum(const float* __restrict__ x, int n, float* __restrict__ out) {
  __shared__ float sh[1024];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < n; i += 1024) acc += x[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

}  // namespace gen
}  // namespace njode
