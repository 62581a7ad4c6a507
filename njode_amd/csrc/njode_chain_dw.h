// Weight gradients of the ODE network behind the wave-per-chain sweeps (njode_chain.h,
// njode_chain_seg.h) -- round 6.  Reference: the autograd of ODEFunc's three linear layers over
// every Euler step (/root/reference/NJODE/models.py:369-377 inside the loop of :430-445).
//
//   dW3 += (dt lam) (x) [a2, 1]     dW2 += delta2 (x) [a1, 1]     dW1 += delta1 (x) [in0, 1]
//
// The sweeps leave everything these three sums need in HBM, one record per (Euler step, path) pair:
// the adjoint lam [H], the state h [H], the hidden activations a1 | a2 and the deltas delta1 | delta2
// as the 64 lanes of the wave held them (lane 16 g + c = unit 4 c + g: njode_dpp.h).  So this kernel
// holds NO weight, draws no mask, computes no transposed product: it is outer products with
// K = the pairs, on the f32 matrix cores, and it reads its MFMA operands STRAIGHT from the records --
// no LDS image, no transposition:
//
//   * k-step s of a tile of 16 pairs is the four pairs p0 + 4 s + g; lane (g, c) loads entries
//     4 c .. 4 c + 3 of pair g's record with ONE global_load_dwordx4 (the 16 lanes of a group read the
//     record's 256 bytes back to back) and has, in the four registers, row / column c of FOUR operand
//     tiles: tile j's row i is record entry 4 i + j.  Which unit that is only matters at the flush
//     (entry e = 16 g' + c' holds unit 4 c' + g'  =>  row 4 g + r of tile j is unit 16 r + 4 j + g,
//     column c of tile j is unit 16 (c mod 4) + 4 j + c / 4).
//   * the input of layer 1 is [tanh h, tanh x, tau, t - tau, (t), 1] and only tanh h (and the bias
//     unit) change from step to step: x and tau are those of the observation in front of the
//     SEGMENT.  The sweeps therefore also leave, per segment, S0 = sum delta1 and S1 = sum delta1
//     (t - tau) over its steps (one more fma per step there), and the x / tau / time columns of dW1
//     are S0 (x) [tanh x, tau] and S1 -- one outer product per SEGMENT instead of one per step, on
//     a few blocks of their own behind the pair blocks of the same launch (the masked 41-dimensional
//     models: 48 instead of 96 matrix instructions per tile for dW1, no x loads, no 41 tanh per
//     lane, and no src_row -> X / t_of_row -> time chain of dependent loads in front of a tile).
//   * a wave owns every accumulator tile (40 for d = H = 41, W = 50: 160 registers) and walks its
//     tiles with the NEXT tile's 44 loads in flight -- two register sets with compile-time indices,
//     the loads of k-step s of the next tile issued behind the products of k-step s of this one, so
//     that every wait the compiler places is a counted one (njode_chain.h's sweep, same reasons).
//   * the four waves of a block sum their tiles through LDS in fixed order: one slab row per block.
//
// What it replaced: k_ode_dw_pairs_mfma (njode_mfma_lockstep.h) on the stored activations, which
// recomputed the three transposed products, staged delta / activation images through LDS and loaded
// src_row -> x per pair: 1.84 ms at 1 000 PhysioNet-shaped paths, 29 us at 100 demo paths
// (profiles/r06_config5_kernels.jsonl, r06_small_batch_kernels.txt).  That kernel still runs when the
// records of the deltas do not fit the budget (KArgs::cdelta null).
#pragma once
#include "njode_mfma.h"
#include "njode_mfma_rows.h"

namespace njode {

template <class C> struct ChainDw {
  static constexpr int H = C::H, W = C::W, D = C::D, IN0 = C::ODE_IN;
  static constexpr int MTH = (H + 15) / 16;          // row tiles of dt lam (natural unit order)
  static constexpr int NTH = (H + 1 + 15) / 16;      // column tiles of [tanh h, 1] (natural order)
  static constexpr int NX = D + 2 + (C::CURT ? 1 : 0);   // tanh x, tau, t - tau, (t)
  static constexpr int NTX = (NX + 15) / 16;
  static constexpr int G3 = 0, G2 = G3 + MTH * 4, G1 = G2 + 16, NP = G1 + 4 * NTH;   // pair role's tiles
  static constexpr int NS = 4 * NTX;                                                 // segment role's
  static constexpr int NRED = NP > NS ? NP : NS;
  static constexpr int EW = 16 * (W % 4) + W / 4;    // record entry of unit W: the bias unit's slot
  // blocks per CU the kernel is compiled for: with H <= 16 (24 accumulator tiles) a wave's tiles + two sets
  // of operands fit 256 registers -- two waves per SIMD, the host then launches up to 512 pair blocks
  static constexpr int BLOCKS_PER_CU = NP <= 24 ? 2 : 1;
  // unit of row 4 g + r / of column c of operand tile j (see above)
  static NJ_DEV int row_unit(int j, int g, int r) { return 16 * r + 4 * j + g; }
  static NJ_DEV int col_unit(int j, int c) { return 16 * (c & 3) + 4 * j + (c >> 2); }
};

// operands of one tile (16 pairs), as loaded: [k-step][operand tile]
template <class C> struct DwOps {
  using T = ChainDw<C>;
  f32x4 a2[4], d2[4], a1[4], d1[4];
  float l3[4][T::MTH], hh[4][T::NTH], dt[4];
};

template <int N> NJ_DEV void dw_block_reduce(f32x4 (&G)[N], float* lds_raw, int wv, int lane) {
  f32x4* red = (f32x4*)lds_raw;
#pragma unroll 1
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wv == w) {
#pragma unroll
      for (int i = 0; i < N; ++i) red[i * 64 + lane] = G[i];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int i = 0; i < N; ++i) G[i] += red[i * 64 + lane];
    }
  }
}

// block `block` of nbp pair blocks + nbs segment blocks
template <class C>
NJ_DEV void ode_dw_stored_body(const KArgs& a, float* lds_raw, int block, int nbp, int nbs) {
  using T = ChainDw<C>;
  using NL = typename C::Ode;
  constexpr int H = C::H, W = C::W, D = C::D, IN0 = C::ODE_IN;
  static_assert(W < 64 && C::NH == 2, "a free lane for the bias unit; two hidden layers");
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float* slab = a.slab + (size_t)block * C::P + C::OFF_ODE;
  float *W1 = slab + NL::woff(0), *b1 = slab + NL::boff(0), *W2 = slab + NL::woff(1),
        *b2 = slab + NL::boff(1), *W3 = slab + NL::woff(2), *b3 = slab + NL::boff(2);

  if (block < nbp) {
    // =========================== pair role ===========================
    const int wave = block * 4 + wv, n_waves = nbp * 4;
    const long long n_pairs = (long long)a.K * a.B;
    const int n_tiles = (int)((n_pairs + 15) / 16);
    const float* const rec = a.chain ? a.lact : a.act;
    f32x4 G[T::NP];
#pragma unroll
    for (int i = 0; i < T::NP; ++i) G[i] = zero4;

    // loads of k-step S of tile `tile` into set o (unconditional: a tile past the end re-reads the last one)
    auto fetch_step = [&](DwOps<C>& o, auto S_, int tile) {
      constexpr int S = decltype(S_)::value;
      const long long p0 = (long long)tile * 16 + 4 * S + g;
      const long long p = p0 < n_pairs ? p0 : n_pairs - 1;
      const float* r = rec + (size_t)p * CHAIN_ACT_FLOATS + 4 * c;
      const float* dl = a.cdelta + (size_t)p * CHAIN_ACT_FLOATS + 4 * c;
      o.a1[S] = *(const f32x4*)r;
      o.a2[S] = *(const f32x4*)(r + 64);
      o.d1[S] = *(const f32x4*)dl;
      o.d2[S] = *(const f32x4*)(dl + 64);
      const float* lm = a.lam_traj + (size_t)p * H;
      const float* hp = a.ltraj + (size_t)p * H;
#pragma unroll
      for (int mt = 0; mt < T::MTH; ++mt) o.l3[S][mt] = lm[16 * mt + c < H ? 16 * mt + c : 0];
#pragma unroll
      for (int nt = 0; nt < T::NTH; ++nt) o.hh[S][nt] = hp[16 * nt + c < H ? 16 * nt + c : 0];
      const unsigned k = (unsigned)p / (unsigned)a.B;
      o.dt[S] = a.step_dt[k];
    };
    // the products of k-step S
    auto consume_step = [&](DwOps<C>& o, auto S_, int tile) {
      constexpr int S = decltype(S_)::value;
      f32x4 a1 = o.a1[S], a2 = o.a2[S], d1 = o.d1[S], d2 = o.d2[S];
      const bool valid = (long long)tile * 16 + 4 * S + g < n_pairs;
      // (a pair past the end of the last tile re-read another pair's records: taken out here)
      const float dt = valid ? o.dt[S] : 0.0f;
      if (c == T::EW / 4) {   // the bias unit of the next layer's input
        a1[T::EW % 4] = 1.0f;
        a2[T::EW % 4] = 1.0f;
      }
      d1 = valid ? d1 : zero4;
      d2 = valid ? d2 : zero4;
      float l3[T::MTH], th[T::NTH];
#pragma unroll
      for (int mt = 0; mt < T::MTH; ++mt) l3[mt] = 16 * mt + c < H ? dt * o.l3[S][mt] : 0.0f;
#pragma unroll
      for (int nt = 0; nt < T::NTH; ++nt) {
        const int hu = 16 * nt + c;
        th[nt] = hu < H ? tanh_f(o.hh[S][nt]) : (hu == H ? 1.0f : 0.0f);
      }
#pragma unroll
      for (int mt = 0; mt < T::MTH; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) G[T::G3 + mt * 4 + j] = mfma4(l3[mt], a2[j], G[T::G3 + mt * 4 + j]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) G[T::G2 + i * 4 + j] = mfma4(d2[i], a1[j], G[T::G2 + i * 4 + j]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int nt = 0; nt < T::NTH; ++nt)
          G[T::G1 + i * T::NTH + nt] = mfma4(d1[i], th[nt], G[T::G1 + i * T::NTH + nt]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    DwOps<C> oA, oB;
    auto fetch_tile = [&](DwOps<C>& o, int tile) {
      fetch_step(o, S0{}, tile);
      fetch_step(o, S1{}, tile);
      fetch_step(o, S2{}, tile);
      fetch_step(o, S3{}, tile);
    };
    // this tile's products, k-step by k-step, each followed by the same k-step's loads of the tile after
    auto tile_body = [&](DwOps<C>& cur, DwOps<C>& nxt, int tile, int tile_n) {
      consume_step(cur, S0{}, tile);
      fetch_step(nxt, S0{}, tile_n);
      consume_step(cur, S1{}, tile);
      fetch_step(nxt, S1{}, tile_n);
      consume_step(cur, S2{}, tile);
      fetch_step(nxt, S2{}, tile_n);
      consume_step(cur, S3{}, tile);
      fetch_step(nxt, S3{}, tile_n);
    };
    if (wave < n_tiles) {
      const int last_t = n_tiles - 1;
      fetch_tile(oA, wave);
      for (int tile = wave;;) {
        int tn = tile + n_waves;
        tile_body(oA, oB, tile, tn < n_tiles ? tn : last_t);
        if (tn >= n_tiles) break;
        tile = tn;
        tn = tile + n_waves;
        tile_body(oB, oA, tile, tn < n_tiles ? tn : last_t);
        if (tn >= n_tiles) break;
        tile = tn;
      }
    }
    dw_block_reduce<T::NP>(G, lds_raw, wv, lane);
    if (wv != 0) return;
    // flush in the parameter layout
#pragma unroll
    for (int mt = 0; mt < T::MTH; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int uo = 16 * mt + 4 * g + r;
        if (uo < H) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ui = T::col_unit(j, c);
            if (ui < W) W3[uo * W + ui] = G[T::G3 + mt * 4 + j][r];
            else if (ui == W) b3[uo] = G[T::G3 + mt * 4 + j][r];
          }
        }
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int uo = T::row_unit(i, g, r);
        if (uo < W) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ui = T::col_unit(j, c);
            if (ui < W) W2[uo * W + ui] = G[T::G2 + i * 4 + j][r];
            else if (ui == W) b2[uo] = G[T::G2 + i * 4 + j][r];
          }
#pragma unroll
          for (int nt = 0; nt < T::NTH; ++nt) {
            const int hu = 16 * nt + c;
            if (hu < H) W1[uo * IN0 + D + hu] = G[T::G1 + i * T::NTH + nt][r];
            else if (hu == H) b1[uo] = G[T::G1 + i * T::NTH + nt][r];
          }
        }
      }
    // (the x / tau / time columns of this row belong to the segment role's rows)
    for (int i = lane; i < W * T::NX; i += 64) {
      const int uo = i / T::NX, xu = i % T::NX;
      W1[uo * IN0 + (xu < D ? xu : H + xu)] = 0.0f;
    }
    return;
  }

  // =========================== segment role ===========================
  {
    const int wave = (block - nbp) * 4 + wv, n_waves = nbs * 4;
    const int n_seg = a.chain ? a.n_obs + a.B : a.n_obs;
    const int n_tiles = (n_seg + 15) / 16;
    f32x4 G[T::NS];
#pragma unroll
    for (int i = 0; i < T::NS; ++i) G[i] = zero4;
    for (int tile = wave; tile < n_tiles; tile += n_waves) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int sg0 = tile * 16 + 4 * s + g;
        const bool valid = sg0 < n_seg;
        const int sg = valid ? sg0 : 0;
        // the observation in front of the segment (-1: the path's start) and its path
        int src, b;
        if (a.chain) {   // lockstep plan: the segment BEHIND row sg; behind the rows, the start segments
          src = sg < a.n_obs ? sg : -1;
          b = sg < a.n_obs ? a.obs_idx[sg] : sg - a.n_obs;
        } else {         // segment plan: the item that ENDS at row sg
          src = a.item_prev[sg];
          b = a.obs_idx[sg];
        }
        const int sv = src >= 0 ? src : 0;
        const float* xp = src >= 0 ? (C::MASKED ? a.y_row + (size_t)sv * C::DO : a.X + (size_t)sv * D)
                                   : a.start_X + (size_t)b * D;
        const float tau = src >= 0 ? a.time_f32[a.t_of_row[sv]] : 0.0f;
        const float* sr = a.cseg + (size_t)sg * CHAIN_ACT_FLOATS + 4 * c;
        f32x4 s0 = *(const f32x4*)sr, s1 = *(const f32x4*)(sr + 64);
        s0 = valid ? s0 : zero4;
        s1 = valid ? s1 : zero4;
        float bx[T::NTX], bt[T::NTX];
#pragma unroll
        for (int nt = 0; nt < T::NTX; ++nt) {
          const int xu = 16 * nt + c;
          const float xv = xp[xu < D ? xu : 0];
          // S0 (x) [tanh x, tau, 0, (tau)]  +  S1 (x) [0, 0, 1, (1)]:  sum delta1 (tau + (t - tau)) = tau S0 + S1
          bx[nt] = xu < D ? tanh_f(xv) : ((xu == D || (C::CURT && xu == D + 2)) ? tau : 0.0f);
          bt[nt] = (xu == D + 1 || (C::CURT && xu == D + 2)) ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int nt = 0; nt < T::NTX; ++nt) {
            G[i * T::NTX + nt] = mfma4(s0[i], bx[nt], G[i * T::NTX + nt]);
            if (16 * nt + 15 >= D + 1 && 16 * nt <= D + 2)   // (the tile that holds the time columns)
              G[i * T::NTX + nt] = mfma4(s1[i], bt[nt], G[i * T::NTX + nt]);
          }
      }
    }
    dw_block_reduce<T::NS>(G, lds_raw, wv, lane);
    // everything of this row's ODE part that is not an x / tau / time column of W1: zero
    for (int i = threadIdx.x; i < NL::SIZE; i += 256) {
      const int iw = i - NL::woff(0);
      bool mine = false;
      if (iw >= 0 && iw < W * IN0) {
        const int col = iw % IN0;
        mine = col < D || col >= H + D;
      }
      if (!mine) slab[i] = 0.0f;
    }
    if (wv != 0) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int uo = T::row_unit(i, g, r);
        if (uo < W) {
#pragma unroll
          for (int nt = 0; nt < T::NTX; ++nt) {
            const int xu = 16 * nt + c;
            if (xu < T::NX) W1[uo * IN0 + (xu < D ? xu : H + xu)] = G[i * T::NTX + nt][r];
          }
        }
      }
  }
}

template <class C>
__global__ void __launch_bounds__(256, ChainDw<C>::BLOCKS_PER_CU) k_ode_dw_stored(KArgs a, int nbp) {
  __shared__ __attribute__((aligned(16))) float lds_raw[ChainDw<C>::NRED * 64 * 4];
  ode_dw_stored_body<C>(a, lds_raw, (int)blockIdx.x, nbp, (int)gridDim.x - nbp);
}
// ... with the encoder's weight-gradient pass of the segment plan (k_encode_rows_bwd_mfma,
// njode_mfma_rows.h) as a third role on the blocks behind: it waits for the same sweep and nothing
// else, and at the reference's batch sizes neither pass fills the chip (B = 100: 15 + 13 us one after
// the other)
template <class C, bool DROP>
__global__ void __launch_bounds__(256, ChainDw<C>::BLOCKS_PER_CU) k_ode_dw_stored_enc(KArgs a, int nbp, int nbs) {
  constexpr int LDS = ChainDw<C>::NRED * 64 * 4 > EncBwdLds<C>::FLOATS ? ChainDw<C>::NRED * 64 * 4 : EncBwdLds<C>::FLOATS;
  __shared__ __attribute__((aligned(16))) float lds_raw[LDS];
  const int block = (int)blockIdx.x;
  if (block < nbp + nbs) ode_dw_stored_body<C>(a, lds_raw, block, nbp, nbs);
  else encode_rows_bwd_body<C, DROP>(a, lds_raw, block - nbp - nbs, (int)gridDim.x - nbp - nbs);
}

}  // namespace njode
