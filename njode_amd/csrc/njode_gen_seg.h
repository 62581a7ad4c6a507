// njode_gen_seg.h -- segment plan of the shape-generic kernel family (njode_gen.h).
//
// Unmasked models: after a jump the state is encoder(X_obs) and does not depend on the state
// before it (models.py:469), so every (path, inter-observation segment) is an independent work
// item, exactly as in the specialised kernels: B paths with n_obs observations
// are n_obs items of ~K B / n_obs Euler steps instead of B serial chains of K steps with three
// network evaluations at every observation time.  An ITEM is named by the observation row it ends
// at; 16 items of (almost) equal length -- the rows sorted by length, descending -- form a tile,
// one workgroup per tile.
//
//   k_gseg_enc       h0 of every observation row and every start value      (row tiles + path tiles)
//   k_gseg_ode_fwd   Euler evolve of every item -> h_end; tails -> hT       (item tiles, tail tiles)
//   k_gseg_mid       readouts before / after the jump, loss terms; training calls: their adjoint
//                    -> lam_end (at the item's end), g_h0 (at the next item's start)  (row tiles)
//   k_gseg_ode_bwd   reverse sweep of every item -> lam_start                (item tiles)
//   k_gseg_enc_bwd   deltas of the encoder evaluations                       (row tiles + path tiles)
//   k_gen_dw         weight gradients: GEMMs over the records, as for the lockstep plan
//
// Records, dropout keys (seed, global path, Euler step of the event, network, layer, unit) and the
// per-chain mathematics are those of k_gen_fwd / k_gen_bwd: the two plans draw the same masks and
// agree to rounding (tests/test_hip_generic.py compares them).
#pragma once
#include "njode_gen.h"

namespace njode {
namespace gen {

struct GSeg {
  const int* order;        // [n_obs] rows sorted by item length, descending
  const int* item_prev;    // [n_obs] previous row of the same path, or -1
  const int* item_next;    // [n_obs] next row of the same path, or -1
  const int* item_kbeg;    // [n_obs] first Euler step of the item
  const int* item_len;     // [n_obs] Euler steps of the item
  const int* t_of_row;     // [n_obs] time index of the row
  const int* first_row;    // [B]
  const int* last_row;     // [B]
  const int* tile_base;    // [NT + 1] first ODE record of an item tile; [NT] = number of records
  const int* tail_order;   // [B] paths sorted by tail length, descending
  const int* k_jump;       // [n_times]
  const int* obs_idx;      // [n_obs]
  int NT;                  // item tiles = row tiles = ceil(n_obs / 16)
  int TB;                  // path tiles = ceil(B / 16)
  int tails;               // the forward also evolves every path from its last row to the end -> hT
  float* h0row;            // [n_obs][H] state after the jump at the row
  float* h0start;          // [B][H]
  float* h_end;            // [n_obs][H] state before the jump at the row
  float* lam_end;          // [n_obs][H]
  float* g_h0;             // [n_obs][H]
  float* lam_start;        // [n_obs][H] adjoint at the START of the item that ends at the row
  float* loss_rows;        // [n_obs]
};

// rows of a [n][W] row-major array <-> tile image [unit][chain]; ids[c] < 0: zeros / not stored
NJ_DEV void load_rows(lfp img, const float* __restrict__ src, const int* ids, int W) {
  for (int e = threadIdx.x; e < 16 * W; e += blockDim.x) {
    const int c = e / W, j = e - c * W, r = ids[c];
    img[j * 16 + c] = r >= 0 ? src[(size_t)r * W + j] : 0.0f;
  }
}
NJ_DEV void store_rows(float* __restrict__ dst, lfp img, const int* ids, int W) {
  for (int e = threadIdx.x; e < 16 * W; e += blockDim.x) {
    const int c = e / W, j = e - c * W, r = ids[c];
    if (r >= 0) dst[(size_t)r * W + j] = img[j * 16 + c];
  }
}

// (obs_idx is the caller's: a value outside [0, B) -- forbidden by the batch layout, reported under
// NJODE_VALIDATE -- must not become an address)
NJ_DEV int seg_path(const GArgs& a, int p) { return p < 0 ? 0 : (p >= a.B ? a.B - 1 : p); }

NJ_DEV void seg_clear_lds(const GArgs& a, lfp smem) {
  const int n = gen_lds_floats(a.img_rows, a.D, a.H, a.DO);
  for (int e = threadIdx.x; e < n; e += blockDim.x) smem[e] = 0.0f;
}

// ---- encoder on observation rows and start values ---------------------------------------------
__global__ void __launch_bounds__(1024) k_gseg_enc(GArgs a, GSeg g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  GLds S;
  S.carve((lfp)smem, a);
  const int tid = threadIdx.x, tile = blockIdx.x;
  seg_clear_lds(a, (lfp)smem);
  __syncthreads();
  const bool start = tile >= g.NT;
  const int i0 = (start ? tile - g.NT : tile) * 16, n = start ? a.B : a.n_obs;
  int* ids = S.rows;
  int* pathI = (int*)(S.misc + 48);
  int* keyI = (int*)(S.misc + 64);
  if (tid < 16) {
    const int i = i0 + tid;
    const bool ok = i < n;
    ids[tid] = ok ? i : -1;
    pathI[tid] = ok ? (start ? i : seg_path(a, g.obs_idx[i])) : 0;
    keyI[tid] = (ok && !start) ? g.k_jump[g.t_of_row[i]] : 0;
  }
  __syncthreads();
  load_rows(S.xr, start ? a.start_X : a.X, ids, a.D);
  __syncthreads();
  enc_input(a, S.img0, S.xr, S.mk);
  __syncthreads();
  const int cch = tid & 15;
  const uint32_t dbase = drop_base(a.dc, a.gid0 + (unsigned long long)pathI[cch],
                                   start ? G_TKEY_START : (uint32_t)keyI[cch], G_NET_ENC);
  float* rec = a.save ? a.rec_enc + (size_t)tile * a.enc.rec_rows * 16 : nullptr;
  lfp out = net_forward(a, a.enc, S.img0, S.img1, rec, a.drop != 0, dbase);
  img_get(S.h, out, a.H);
  __syncthreads();
  enc_residual(a, S.h, S.xr);
  __syncthreads();
  store_rows(start ? g.h0start : g.h0row, S.h, ids, a.H);
}

// ---- items: what a tile's chains are ------------------------------------------------------------
struct SegTile {
  int *row, *prev, *path, *kbeg, *len;   // LDS, 16 each
  lfp tau, tcur, dtc;
  int maxlen;
};
NJ_DEV void seg_tile_setup(const GArgs& a, const GSeg& g, GLds& S, SegTile& t, int tile, bool tail) {
  t.row = S.rows;
  t.prev = (int*)(S.misc + 16);
  t.path = (int*)(S.misc + 32);
  t.kbeg = (int*)(S.misc + 48);
  t.len = (int*)(S.misc + 64);
  t.tau = S.misc;
  t.tcur = S.misc + 80;
  t.dtc = S.misc + 96;                   // two slots of 16
  const int tid = threadIdx.x;
  if (tid < 16) {
    const int i = tile * 16 + tid;
    int row = -1, prev = -1, path = 0, kbeg = 0, len = 0;
    float tau = 0.0f;
    if (!tail) {
      if (i < a.n_obs) {
        row = g.order[i];
        prev = g.item_prev[row];
        path = seg_path(a, g.obs_idx[row]);
        kbeg = g.item_kbeg[row];
        len = g.item_len[row];
      }
    } else if (i < a.B) {
      path = g.tail_order[i];
      row = path;                          // (tails: the id the result is stored under)
      prev = g.last_row[path];
      kbeg = prev >= 0 ? g.k_jump[g.t_of_row[prev]] : 0;
      len = a.K - kbeg;
    }
    if (prev >= 0) tau = a.time_f32[g.t_of_row[prev]];
    if (len < 0) len = 0;
    t.row[tid] = row;
    t.prev[tid] = prev;
    t.path[tid] = path;
    t.kbeg[tid] = kbeg;
    t.len[tid] = row >= 0 ? len : 0;
    t.tau[tid] = tau;
  }
  __syncthreads();
  int m = 0;
#pragma unroll
  for (int c = 0; c < 16; ++c) m = t.len[c] > m ? t.len[c] : m;
  t.maxlen = m;
}
// time and step size of every chain at local step s (inactive chains: dt = 0), fetched into
// registers of the first 16 threads one step before they are published in LDS; dt goes to one
// of two slots (s & 1) so that the values of step s + 1 can be written while step s still reads
// its own
struct SegClock { float t, dt; };
NJ_DEV SegClock seg_clock_fetch(const GArgs& a, const SegTile& t, int s) {
  SegClock c{0.0f, 0.0f};
  const int tid = threadIdx.x;
  if (tid < 16 && s >= 0 && a.K > 0) {
    int k = t.kbeg[tid] + s;
    const bool on = s < t.len[tid];
    if (k > a.K - 1) k = a.K - 1;
    if (k < 0) k = 0;
    c.t = a.step_t[k];
    c.dt = on ? a.step_dt[k] : 0.0f;
  }
  return c;
}
NJ_DEV void seg_clock_put(const SegTile& t, int s, const SegClock c) {
  const int tid = threadIdx.x;
  if (tid < 16 && s >= 0) {
    t.tcur[tid] = c.t;
    t.dtc[(s & 1) * 16 + tid] = c.dt;
  }
}
// ODE input with a time per chain (ode_input of njode_gen.h takes one time for the tile)
NJ_DEV void seg_ode_input(const GArgs& a, lfp in, lfp tx, lfp h, lfp tau, lfp tcur) {
  const int n = (a.D + a.H) * 16;
  for (int e = threadIdx.x; e < n; e += blockDim.x)
    in[pix(e)] = e < a.D * 16 ? tx[e] : tanh_acc(h[e - a.D * 16]);
  if (threadIdx.x < 16) {
    const int c = threadIdx.x;
    const float ta = tau[c], td = tcur[c] - ta;
    in[pix((a.D + a.H) * 16 + c)] = ta;
    in[pix((a.D + a.H + 1) * 16 + c)] = td;
    if (a.curt) in[pix((a.D + a.H + 2) * 16 + c)] = ta + td;
    in[pix(a.IN0 * 16 + c)] = 1.0f;                          // bias unit
  }
}

// ---- Euler evolve of the items (and of the tails) -------------------------------------------------
__global__ void __launch_bounds__(1024) k_gseg_ode_fwd(GArgs a, GSeg g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  GLds S;
  S.carve((lfp)smem, a);
  const int tid = threadIdx.x, nth = blockDim.x;
  seg_clear_lds(a, (lfp)smem);
  __syncthreads();
  const bool tail = (int)blockIdx.x >= g.NT;
  const int tile = tail ? blockIdx.x - g.NT : blockIdx.x;
  SegTile t;
  seg_tile_setup(a, g, S, t, tile, tail);
  // start state, last observation
  for (int e = tid; e < 16 * a.H; e += nth) {
    const int c = e / a.H, j = e - c * a.H;
    float v = 0.0f;
    if (t.row[c] >= 0) v = t.prev[c] >= 0 ? g.h0row[(size_t)t.prev[c] * a.H + j] : g.h0start[(size_t)t.path[c] * a.H + j];
    S.h[j * 16 + c] = v;
  }
  for (int e = tid; e < 16 * a.D; e += nth) {
    const int c = e / a.D, q = e - c * a.D;
    float v = 0.0f;
    if (t.row[c] >= 0) v = t.prev[c] >= 0 ? a.X[(size_t)t.prev[c] * a.D + q] : a.start_X[(size_t)t.path[c] * a.D + q];
    S.tx[q * 16 + c] = tanh_acc(v);
  }
  seg_clock_put(t, 0, seg_clock_fetch(a, t, 0));
  SegClock nclk = seg_clock_fetch(a, t, 1);
  __syncthreads();
  const int cch = tid & 15;
  const unsigned long long gidc = a.gid0 + (unsigned long long)t.path[cch];
  const uint32_t kb = (uint32_t)t.kbeg[cch];
  const bool save = a.save && !tail;
  const size_t rbase = save ? (size_t)g.tile_base[tile] : 0;
  for (int s = 0; s < t.maxlen; ++s) {
#ifdef NJ_GEN_STAMPS
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && tid == 0) g_gen_stamps[15] += 1;
#endif
    if (!ABL(a.dbg, 16)) seg_ode_input(a, S.img0, S.tx, S.h, t.tau, t.tcur);
    if (!ABL(a.dbg, 64)) __syncthreads();
    GSTAMP(0, t0);
    seg_clock_put(t, s + 1, nclk);   // (tcur of step s was consumed above; dt goes to the other slot)
    nclk = seg_clock_fetch(a, t, s + 2);
    float* rec = save ? a.rec_ode + (rbase + s) * a.ode.rec_rows * 16 : nullptr;
    lfp out = net_forward(a, a.ode, S.img0, S.img1, rec, a.drop != 0,
                          drop_base(a.dc, gidc, kb + (uint32_t)s, G_NET_ODE));
    GSTAMP(1, t0);
    lfp dts = t.dtc + (s & 1) * 16;
    if (!ABL(a.dbg, 16))
      for (int e = tid; e < a.H * 16; e += nth) {
        const float dt = dts[e & 15];
        if (dt != 0.0f) S.h[e] = fmaf(dt, out[pix(e)], S.h[e]);
      }
    if (!ABL(a.dbg, 64)) __syncthreads();
    GSTAMP(2, t0);
  }
  store_rows(tail ? a.hT : g.h_end, S.h, t.row, a.H);
}

// ---- readouts at the observation rows, loss; training calls: their adjoint -------------------------
__global__ void __launch_bounds__(1024) k_gseg_mid(GArgs a, GSeg g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  GLds S;
  S.carve((lfp)smem, a);
  const int tid = threadIdx.x, nth = blockDim.x, tile = blockIdx.x;
  seg_clear_lds(a, (lfp)smem);
  __syncthreads();
  int* ids = S.rows;
  int* pathI = (int*)(S.misc + 48);
  int* keyI = (int*)(S.misc + 64);
  lfp lossacc = S.misc + 16, scale = S.misc + 32;
  if (tid < 16) {
    const int r = tile * 16 + tid;
    const bool ok = r < a.n_obs;
    const int path = ok ? seg_path(a, g.obs_idx[r]) : 0;
    ids[tid] = ok ? r : -1;
    pathI[tid] = path;
    keyI[tid] = ok ? g.k_jump[g.t_of_row[r]] : 0;
    scale[tid] = ok ? a.inv_batch / (float)a.n_obs_ot[path] : 0.0f;
  }
  __syncthreads();
  load_rows(S.h, g.h_end, ids, a.H);
  load_rows(S.hn, g.h0row, ids, a.H);
  load_rows(S.xr, a.X, ids, a.D);
  __syncthreads();
  const int cch = tid & 15;
  const unsigned long long gidc = a.gid0 + (unsigned long long)pathI[cch];
  const uint32_t tk = (uint32_t)keyI[cch];
  float* rec_b = a.save ? a.rec_dec + ((size_t)tile * 2 + 0) * a.dec.rec_rows * 16 : nullptr;
  float* rec_y = a.save ? a.rec_dec + ((size_t)tile * 2 + 1) * a.dec.rec_rows * 16 : nullptr;
  // y_bj = readout(state before the jump)
  dec_input(a, S.img0, S.h);
  __syncthreads();
  {
    lfp out = net_forward(a, a.dec, S.img0, S.img1, rec_b, a.drop != 0, drop_base(a.dc, gidc, tk, G_NET_DEC_BJ));
    img_get(S.ybj, out, a.DO);
    __syncthreads();
    dec_residual(a, S.ybj, S.h);
    __syncthreads();
  }
  // y = readout(state after the jump)
  dec_input(a, S.img0, S.hn);
  __syncthreads();
  {
    lfp out = net_forward(a, a.dec, S.img0, S.img1, rec_y, a.drop != 0, drop_base(a.dc, gidc, tk, G_NET_DEC));
    img_get(S.y, out, a.DO);
    __syncthreads();
    dec_residual(a, S.y, S.hn);
    __syncthreads();
  }
  lfp dy = S.xin, dybj = S.tx;
  loss_tile(a, S.xr, S.mk, S.y, S.ybj, ids, scale, lossacc, a.save ? dy : nullptr, a.save ? dybj : nullptr);
  __syncthreads();
  if (tid < 16 && ids[tid] >= 0) g.loss_rows[ids[tid]] = lossacc[tid];
  if (!a.save) return;
  const bool drop = a.drop != 0;
  // adjoint of y = readout(h0row): gradient at the start of the next item
  img_put(S.img0, dy, a.DO);
  __syncthreads();
  {
    lfp din = net_backward(a, a.dec, S.img0, S.img1, rec_y, drop, true);
    const float* th = rec_y + (size_t)a.dec.l[0].a_row * 16;
    for (int e = tid; e < a.H * 16; e += nth) {
      const int j = e >> 4, c = e & 15;
      const float tv = th[e];
      float v = din[pix(e)] * (1.0f - tv * tv);
      if (a.dec_case == 1) {
        for (int q = j; q < a.DO; q += a.H) v += dy[q * 16 + c];
      } else if (a.dec_case == 2) {
        v += dy[(j % a.DO) * 16 + c] * (1.0f / a.dec_mult);
      }
      S.hn[e] = v;
    }
    __syncthreads();
    store_rows(g.g_h0, S.hn, ids, a.H);
  }
  // adjoint of y_bj = readout(h_end): gradient at the end of the item
  img_put(S.img0, dybj, a.DO);
  __syncthreads();
  {
    lfp din = net_backward(a, a.dec, S.img0, S.img1, rec_b, drop, true);
    const float* th = rec_b + (size_t)a.dec.l[0].a_row * 16;
    for (int e = tid; e < a.H * 16; e += nth) {
      const int j = e >> 4, c = e & 15;
      const float tv = th[e];
      float v = din[pix(e)] * (1.0f - tv * tv);
      if (a.dec_case == 1) {
        for (int q = j; q < a.DO; q += a.H) v += dybj[q * 16 + c];
      } else if (a.dec_case == 2) {
        v += dybj[(j % a.DO) * 16 + c] * (1.0f / a.dec_mult);
      }
      S.h[e] = v;
    }
    __syncthreads();
    store_rows(g.lam_end, S.h, ids, a.H);
  }
}

// ---- reverse sweep of the items -----------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_gseg_ode_bwd(GArgs a, GSeg g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  GLds S;
  S.carve((lfp)smem, a);
  const int tid = threadIdx.x, nth = blockDim.x, tile = blockIdx.x;
  seg_clear_lds(a, (lfp)smem);
  __syncthreads();
  SegTile t;
  seg_tile_setup(a, g, S, t, tile, false);
  lfp lam = S.h;
  load_rows(lam, g.lam_end, t.row, a.H);
  const bool drop = a.drop != 0;
  const size_t rbase = (size_t)g.tile_base[tile];
  __syncthreads();
  seg_clock_put(t, t.maxlen - 1, seg_clock_fetch(a, t, t.maxlen - 1));
  SegClock nclk = seg_clock_fetch(a, t, t.maxlen - 2);
  __syncthreads();
  for (int s = t.maxlen - 1; s >= 0; --s) {
    float* rec = a.rec_ode + (rbase + s) * a.ode.rec_rows * 16;
    lfp dts = t.dtc + (s & 1) * 16;
    for (int e = tid; e < a.H * 16; e += nth) S.img0[pix(e)] = dts[e & 15] * lam[e];
    __syncthreads();
    seg_clock_put(t, s - 1, nclk);
    nclk = seg_clock_fetch(a, t, s - 2);
    lfp din = net_backward(a, a.ode, S.img0, S.img1, rec, drop, true);
    const float* in0 = rec + (size_t)a.ode.l[0].a_row * 16;
    for (int e = tid; e < a.H * 16; e += nth) {
      const float th = in0[a.D * 16 + e];
      lam[e] = fmaf(din[pix(a.D * 16 + e)], 1.0f - th * th, lam[e]);
    }
    __syncthreads();
  }
  store_rows(g.lam_start, lam, t.row, a.H);
}

// ---- deltas of the encoder evaluations -----------------------------------------------------------
__global__ void __launch_bounds__(1024) k_gseg_enc_bwd(GArgs a, GSeg g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  GLds S;
  S.carve((lfp)smem, a);
  const int tid = threadIdx.x, nth = blockDim.x, tile = blockIdx.x;
  seg_clear_lds(a, (lfp)smem);
  __syncthreads();
  const bool start = tile >= g.NT;
  const int i0 = (start ? tile - g.NT : tile) * 16, n = start ? a.B : a.n_obs;
  int* ids = S.rows;
  int* nxt = (int*)(S.misc + 48);
  if (tid < 16) {
    const int i = i0 + tid;
    const bool ok = i < n;
    ids[tid] = (ok && !start) ? i : -1;
    nxt[tid] = ok ? (start ? g.first_row[i] : g.item_next[i]) : -1;
  }
  __syncthreads();
  for (int e = tid; e < 16 * a.H; e += nth) {
    const int c = e / a.H, j = e - c * a.H;
    float v = 0.0f;
    if (ids[c] >= 0) v = g.g_h0[(size_t)ids[c] * a.H + j];
    if (nxt[c] >= 0) v += g.lam_start[(size_t)nxt[c] * a.H + j];
    S.img0[pix(j * 16 + c)] = v;
  }
  __syncthreads();
  float* rec = a.rec_enc + (size_t)tile * a.enc.rec_rows * 16;
  (void)net_backward(a, a.enc, S.img0, S.img1, rec, a.drop != 0, false);
}

// ---- plan --------------------------------------------------------------------------------------
// every path walks its column of the dense [time][path] -> row matrix in time order
__global__ void k_gseg_link(int B, int n_times, int K, const int* __restrict__ dense,
                            const int* __restrict__ k_jump, int* __restrict__ t_of_row,
                            int* __restrict__ item_prev, int* __restrict__ item_next,
                            int* __restrict__ item_kbeg, int* __restrict__ item_len,
                            unsigned* __restrict__ key, int* __restrict__ first_row,
                            int* __restrict__ last_row, unsigned* __restrict__ tail_key,
                            int* __restrict__ iota_b) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int prev = -1, kprev = 0;
  constexpr int CH = 16;
  for (int i0 = 0; i0 < n_times; i0 += CH) {
    int rr[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) rr[q] = i0 + q < n_times ? dense[(size_t)(i0 + q) * B + b] : -1;
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int r = rr[q];
      if (r < 0) continue;
      const int kend = k_jump[i0 + q];
      int len = kend - kprev;
      if (len < 0) len = 0;
      if (len > K) len = K;
      t_of_row[r] = i0 + q;
      item_prev[r] = prev;
      if (prev >= 0) item_next[prev] = r; else first_row[b] = r;
      item_kbeg[r] = kprev;
      item_len[r] = len;
      key[r] = (unsigned)(K - len);        // ascending key == descending length
      prev = r;
      kprev = kend;
    }
  }
  if (prev >= 0) { item_next[prev] = -1; last_row[b] = prev; }
  else { first_row[b] = -1; last_row[b] = -1; }
  int kt = kprev < 0 ? 0 : (kprev > K ? K : kprev);
  tail_key[b] = (unsigned)kt;              // tail length K - kprev, descending
  iota_b[b] = b;
}
// row numbers for the sort, and a harmless item (no steps, no neighbours) for every row: a row the
// column walk does not reach -- a path listed twice in one time slice, which the batch layout
// forbids -- then costs its observation, not memory safety
__global__ void k_gseg_init(int n, int K, int* __restrict__ iota, int* __restrict__ t_of_row,
                            int* __restrict__ item_prev, int* __restrict__ item_next,
                            int* __restrict__ item_kbeg, int* __restrict__ item_len,
                            unsigned* __restrict__ key) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  iota[i] = i;
  t_of_row[i] = 0;
  item_prev[i] = -1;
  item_next[i] = -1;
  item_kbeg[i] = 0;
  item_len[i] = 0;
  key[i] = (unsigned)K;
}
// first ODE record of every item tile: exclusive scan of the tiles' lengths (one workgroup)
__global__ void __launch_bounds__(1024) k_gseg_tiles(const int* __restrict__ order,
                                                     const int* __restrict__ item_len, int n_obs,
                                                     int NT, int* __restrict__ tile_base) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (NT + 1023) / 1024;
  const int lo = tid * per, hi = lo + per < NT ? lo + per : NT;
  int s = 0;
  for (int t = lo; t < hi; ++t) s += item_len[order[t * 16]];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = acc; acc += v; }
    tile_base[NT] = acc;
  }
  __syncthreads();
  int acc = part[tid];
  for (int t = lo; t < hi; ++t) { tile_base[t] = acc; acc += item_len[order[t * 16]]; }
}

}  // namespace gen
}  // namespace njode
