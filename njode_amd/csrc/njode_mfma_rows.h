// njode_mfma_rows.h -- the per-observation-row kernels of the segment plan (encoder,
// readout + loss, and their backward) on the f32 matrix cores.  Same conventions as
// njode_mfma.h: 16 rows per wave, vectors in D-layout (lane (g, c) holds unit 4q + g of
// row c), weights resident in registers as A-fragments, dW through LDS images.
#pragma once
#include "njode_mfma.h"

namespace njode {

// shape of one 2-hidden-layer network on the matrix cores
template <int IN_, int OUT_, int W_> struct MS {
  static constexpr int IN = IN_, OUT = OUT_, W = W_;
  static constexpr int Q0 = (IN + 1 + 3) / 4, Q1 = (W + 1 + 3) / 4;
  static constexpr int QO = (OUT + 3) / 4, QW = (W + 3) / 4, QI = (IN + 3) / 4;
  static constexpr int MT1 = (W + 15) / 16, MTO = (OUT + 15) / 16, MTI = (IN + 15) / 16;
  static constexpr int NT1 = (W + 1 + 15) / 16, NT0 = (IN + 1 + 15) / 16;
  static constexpr int F1 = 0, F2 = F1 + MT1 * Q0, F3 = F2 + MT1 * Q1, NFWD = F3 + MTO * Q1;
  static constexpr int B3 = NFWD, B2 = B3 + MT1 * QO, B1 = B2 + MT1 * QW, NALL = B1 + MTI * QW;
  // edge rows of the W-unit layers (njode_mfma.h): tile ET stands for ER rows on 4x4x1; dW2 is
  // then held as G2M x G2N tiles of 16x16 + two edge accumulators (dw_accumulate_edge)
  static constexpr int ET = EdgeRows<W>::TILE, ER = EdgeRows<W>::R;
  static constexpr int G2M = ER ? ET : MT1, G2N = ER ? ET : NT1;
};

// A-fragments of all six products of one network (inputs in natural order)
template <class NL, class S>
NJ_DEV void pack_net_value(const float* __restrict__ Pn, float* __restrict__ frag, int idx) {
  if (idx >= S::NALL * 64) return;
  const int f = idx >> 6, l = idx & 63, g = l >> 4, c = l & 15;
  const float *W1 = Pn + NL::woff(0), *b1 = Pn + NL::boff(0), *W2 = Pn + NL::woff(1),
              *b2 = Pn + NL::boff(1), *W3 = Pn + NL::woff(2), *b3 = Pn + NL::boff(2);
  float v = 0.0f;
  if (f < S::F2) {
    const int mt = f / S::Q0, q = f % S::Q0, uo = row_unit(mt, c), ui = 4 * q + g;
    if (uo < S::W) v = ui < S::IN ? W1[uo * S::IN + ui] : (ui == S::IN ? b1[uo] : 0.0f);
  } else if (f < S::F3) {
    const int mt = (f - S::F2) / S::Q1, q = (f - S::F2) % S::Q1, uo = row_unit(mt, c), ui = 4 * q + g;
    if (uo < S::W) v = ui < S::W ? W2[uo * S::W + ui] : (ui == S::W ? b2[uo] : 0.0f);
  } else if (f < S::NFWD) {
    const int mt = (f - S::F3) / S::Q1, q = (f - S::F3) % S::Q1, uo = row_unit(mt, c), ui = 4 * q + g;
    if (uo < S::OUT) v = ui < S::W ? W3[uo * S::W + ui] : (ui == S::W ? b3[uo] : 0.0f);
  } else if (f < S::B2) {
    const int mt = (f - S::B3) / S::QO, q = (f - S::B3) % S::QO, ui = row_unit(mt, c), uo = 4 * q + g;
    if (ui < S::W && uo < S::OUT) v = W3[uo * S::W + ui];
  } else if (f < S::B1) {
    const int mt = (f - S::B2) / S::QW, q = (f - S::B2) % S::QW, ui = row_unit(mt, c), uo = 4 * q + g;
    if (ui < S::W && uo < S::W) v = W2[uo * S::W + ui];
  } else {
    const int mt = (f - S::B1) / S::QW, q = (f - S::B1) % S::QW, ui = row_unit(mt, c), uo = 4 * q + g;
    if (ui < S::IN && uo < S::W) v = W1[uo * S::IN + ui];
  }
  frag[idx] = v;
}
template <class NL, class S>
__global__ void k_pack_net(const float* __restrict__ Pn, float* __restrict__ frag) {
  pack_net_value<NL, S>(Pn, frag, blockIdx.x * blockDim.x + threadIdx.x);
}

// Fragment providers.  Register sets (forward kernels: ~70 VGPRs) and LDS views
// (gradient kernels: the fragments of one 256-thread block live once in LDS, so each wave
// stays under 256 registers and two waves fit per SIMD; fragment reads are laundered per
// evaluation so the compiler does not hoist them back into registers).
template <class S> struct FwdFrags {
  float A1[S::MT1][S::Q0], A2[S::MT1][S::Q1], A3[S::MTO][S::Q1];
  NJ_DEV void load(const float* frag, int lane) {
#pragma unroll
    for (int mt = 0; mt < S::MT1; ++mt) {
      const int l = mt == S::ET ? edge_lane(lane) : lane;   // (edge tile: its rows' 4x4x1 operands)
#pragma unroll
      for (int q = 0; q < S::Q0; ++q) A1[mt][q] = frag[(S::F1 + mt * S::Q0 + q) * 64 + l];
#pragma unroll
      for (int q = 0; q < S::Q1; ++q) A2[mt][q] = frag[(S::F2 + mt * S::Q1 + q) * 64 + l];
    }
    // (one output unit: its weight row W3[0][4q + g], the fragment's row 0, see mnet_fwd)
    const int l3 = S::OUT == 1 ? (lane & 48) : lane;
#pragma unroll
    for (int mt = 0; mt < S::MTO; ++mt)
#pragma unroll
      for (int q = 0; q < S::Q1; ++q) A3[mt][q] = frag[(S::F3 + mt * S::Q1 + q) * 64 + l3];
  }
  NJ_DEV float r3(int q) const { return A3[0][q]; }
  NJ_DEV void begin() const {}
  NJ_DEV float a1(int mt, int q) const { return A1[mt][q]; }
  NJ_DEV float a2(int mt, int q) const { return A2[mt][q]; }
  NJ_DEV float a3(int mt, int q) const { return A3[mt][q]; }
  NJ_DEV float e1(int q) const { return A1[S::ET < 0 ? 0 : S::ET][q]; }
  NJ_DEV float e2(int q) const { return A2[S::ET < 0 ? 0 : S::ET][q]; }
};
template <class S> struct LdsFrags {
  lfp base;   // this lane's column of the block's fragment image: base[f * 64]
  lfp cur;
  int de;     // edge rows: their 4x4x1 operands are the edge tile's vectors at lane + de
  int dr;     // one output unit: its weight row = row 0 of the output fragment, at lane 16 g
  NJ_DEV void init(lfp img, int lane) {
    base = img + lane;
    cur = base;
    de = edge_lane(lane) - lane;
    dr = (lane & 48) - lane;
  }
  NJ_DEV float r3(int q) const { return cur[(S::F3 + q) * 64 + dr]; }
  // copy all S::NALL fragment vectors of the network from global into LDS (whole block)
  static NJ_DEV void stage(lfp img, const float* frag, int tid, int nthreads) {
    for (int i = tid; i < S::NALL * 64; i += nthreads) img[i] = frag[i];
  }
  NJ_DEV void begin() {
    unsigned v = (unsigned)(unsigned long long)base;
    asm volatile("" : "+v"(v));
    cur = (lfp)(unsigned long long)v;
  }
  NJ_DEV float a1(int mt, int q) const { return cur[(S::F1 + mt * S::Q0 + q) * 64]; }
  NJ_DEV float a2(int mt, int q) const { return cur[(S::F2 + mt * S::Q1 + q) * 64]; }
  NJ_DEV float a3(int mt, int q) const { return cur[(S::F3 + mt * S::Q1 + q) * 64]; }
  NJ_DEV float b3(int mt, int q) const { return cur[(S::B3 + mt * S::QO + q) * 64]; }
  NJ_DEV float b2(int mt, int q) const { return cur[(S::B2 + mt * S::QW + q) * 64]; }
  NJ_DEV float b1(int mt, int q) const { return cur[(S::B1 + mt * S::QW + q) * 64]; }
  NJ_DEV float e1(int q) const { return cur[(S::F1 + S::ET * S::Q0 + q) * 64 + de]; }
  NJ_DEV float e2(int q) const { return cur[(S::F2 + S::ET * S::Q1 + q) * 64 + de]; }
  NJ_DEV float eb3(int q) const { return cur[(S::B3 + S::ET * S::QO + q) * 64 + de]; }
  NJ_DEV float eb2(int q) const { return cur[(S::B2 + S::ET * S::QW + q) * 64 + de]; }
};
template <class S> struct GradTiles {
  f32x4 G3[S::MTO][S::NT1], G2[S::G2M][S::G2N], GM[2], GN[2], G1[S::MT1][S::NT0];
  NJ_DEV void zero() {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    GM[0] = GM[1] = GN[0] = GN[1] = z;
#pragma unroll
    for (int i = 0; i < S::MTO; ++i)
#pragma unroll
      for (int n = 0; n < S::NT1; ++n) G3[i][n] = z;
#pragma unroll
    for (int i = 0; i < S::G2M; ++i)
#pragma unroll
      for (int n = 0; n < S::G2N; ++n) G2[i][n] = z;
#pragma unroll
    for (int i = 0; i < S::MT1; ++i) {
#pragma unroll
      for (int n = 0; n < S::NT0; ++n) G1[i][n] = z;
    }
  }
  static constexpr int NG = S::MTO * S::NT1 + S::G2M * S::G2N + (S::ER ? 2 : 0) + S::MT1 * S::NT0;
  template <class F> NJ_DEV void for_tiles(F f) {
    int i = 0;
#pragma unroll
    for (int mt = 0; mt < S::MTO; ++mt)
#pragma unroll
      for (int nt = 0; nt < S::NT1; ++nt) f(G3[mt][nt], i++);
#pragma unroll
    for (int mt = 0; mt < S::G2M; ++mt)
#pragma unroll
      for (int nt = 0; nt < S::G2N; ++nt) f(G2[mt][nt], i++);
    if constexpr (S::ER != 0) {
      f(GM[0], i++);
      f(GN[0], i++);
    }
#pragma unroll
    for (int mt = 0; mt < S::MT1; ++mt)
#pragma unroll
      for (int nt = 0; nt < S::NT0; ++nt) f(G1[mt][nt], i++);
  }
  // Sum the four waves' tiles of a 256-thread block into wave 0's (fixed order:
  // deterministic), through `lds` (>= 3 * NG * 256 floats, free by now): the block then
  // stores ONE slab row.  Returns true for the wave that holds the sum.
  NJ_DEV bool reduce_block(lfp lds, int wv, int lane) {
    f32x4 __attribute__((address_space(3)))* red = (f32x4 __attribute__((address_space(3)))*)lds;
    if constexpr (S::ER != 0) {   // (two accumulators each, independent 4x4x1 chains: flush adds them)
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      GM[0] += GM[1];
      GN[0] += GN[1];
      GM[1] = GN[1] = z;
    }
    __syncthreads();
    if (wv > 0) for_tiles([&](f32x4& t, int i) { red[((wv - 1) * NG + i) * 64 + lane] = t; });
    __syncthreads();
    if (wv != 0) return false;
    for_tiles([&](f32x4& t, int i) {
      t += red[(0 * NG + i) * 64 + lane];
      t += red[(1 * NG + i) * 64 + lane];
      t += red[(2 * NG + i) * 64 + lane];
    });
    return true;
  }
  // store into a slab laid out like the network's parameters (NL offsets)
  template <class NL> NJ_DEV void flush(float* slab, int g, int c) const {
    float *W1 = slab + NL::woff(0), *b1 = slab + NL::boff(0), *W2 = slab + NL::woff(1),
          *b2 = slab + NL::boff(1), *W3 = slab + NL::woff(2), *b3 = slab + NL::boff(2);
#pragma unroll
    for (int mt = 0; mt < S::G2M; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int uo = 16 * mt + 4 * g + r;
        if (uo < S::W) {
#pragma unroll
          for (int nt = 0; nt < S::G2N; ++nt) {
            const int ui = 16 * nt + c;
            if (ui < S::W) W2[uo * S::W + ui] = G2[mt][nt][r];
            else if (ui == S::W) b2[uo] = G2[mt][nt][r];
          }
        }
      }
    if constexpr (S::ER != 0) {
      // GM: register i of lane l = dW2[16 ET + i][l]; GN: = dW2[l][16 ET + i] for l < 16 ET
      const int lane = 16 * g + c;
      const f32x4 gm = GM[0] + GM[1], gn = GN[0] + GN[1];
#pragma unroll
      for (int i = 0; i < S::ER; ++i) {
        const int uo = 16 * S::ET + i;
        if (lane < S::W) W2[uo * S::W + lane] = gm[i];
        else if (lane == S::W) b2[uo] = gm[i];
      }
      if (lane < 16 * S::ET) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ui = 16 * S::ET + i;
          if (ui < S::W) W2[lane * S::W + ui] = gn[i];
          else if (ui == S::W) b2[lane] = gn[i];
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < S::MT1; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int uo = 16 * mt + 4 * g + r;
        if (uo < S::W) {
#pragma unroll
          for (int nt = 0; nt < S::NT0; ++nt) {
            const int ui = 16 * nt + c;
            if (ui < S::IN) W1[uo * S::IN + ui] = G1[mt][nt][r];
            else if (ui == S::IN) b1[uo] = G1[mt][nt][r];
          }
        }
      }
    if constexpr (S::OUT == 1) {
      // (mnet_bwd's one-output form: element q of G3[0] is this lane's chain's part of
      // dW3[0][4q + g]; the 16 chains of a lane group are summed here, once per block)
#pragma unroll
      for (int q = 0; q < S::Q1; ++q) {
        float v = G3[0][q / 4][q % 4];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
        const int ui = 4 * q + g;
        if (c == 0) {
          if (ui < S::W) W3[ui] = v;
          else if (ui == S::W) b3[0] = v;
        }
      }
      return;
    }
#pragma unroll
    for (int mt = 0; mt < S::MTO; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int uo = 16 * mt + 4 * g + r;
        if (uo < S::OUT) {
#pragma unroll
          for (int nt = 0; nt < S::NT1; ++nt) {
            const int ui = 16 * nt + c;
            if (ui < S::W) W3[uo * S::W + ui] = G3[mt][nt][r];
            else if (ui == S::W) b3[uo] = G3[mt][nt][r];
          }
        }
      }
  }
};

template <class S, int ACT, bool DROP, class FP>
NJ_DEV void mnet_fwd(FP& F, const float (&b0)[S::Q0], float (&a1)[S::Q1],
                     float (&a2)[S::Q1], f32x4 (&out)[S::MTO], uint32_t k1, uint32_t k2,
                     float inv_keep, int g) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[S::MT1];
  F.begin();
#pragma unroll
  for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = z;
#pragma unroll
  for (int q = 0; q < S::Q0; ++q)
#pragma unroll
    for (int mt = 0; mt < S::MT1; ++mt)
      acc[mt] = mt == S::ET ? mfma1(F.e1(q), b0[q], acc[mt]) : mfma4(F.a1(mt, q), b0[q], acc[mt]);
  if constexpr (S::ER != 0) acc[S::ET] = edge_tile<S::ER>(acc[S::ET], g);
  hidden_from_acc_g<S::MT1, S::Q1, S::W, ACT, DROP>(acc, a1, k1, inv_keep, g);
#pragma unroll
  for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = z;
#pragma unroll
  for (int q = 0; q < S::Q1; ++q)
#pragma unroll
    for (int mt = 0; mt < S::MT1; ++mt)
      acc[mt] = mt == S::ET ? mfma1(F.e2(q), a1[q], acc[mt]) : mfma4(F.a2(mt, q), a1[q], acc[mt]);
  if constexpr (S::ER != 0) acc[S::ET] = edge_tile<S::ER>(acc[S::ET], g);
  hidden_from_acc_g<S::MT1, S::Q1, S::W, ACT, DROP>(acc, a2, k2, inv_keep, g);
  if constexpr (S::OUT == 1) {
    // One output unit (the 1-d models' readout): a 16-row tile per k-step for one row is 13
    // MFMAs = 416 cycles; on the vector pipe it is 13 FMAs with the weight row W3[0][4q + g] + the
    // sum over the four lane groups = ~70.  EVERY lane gets the value (register 0; D-layout asks
    // for it in lane group 0 only, the other groups' units 1 - 3 do not exist and are never read).
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int q = 0; q < S::Q1; q += 2) {
      p0 = fmaf(F.r3(q), a2[q], p0);
      if (q + 1 < S::Q1) p1 = fmaf(F.r3(q + 1), a2[q + 1], p1);
    }
    const float p = p0 + p1;
    out[0] = f32x4{edge_reduce(p, p), 0.0f, 0.0f, 0.0f};
    return;
  }
#pragma unroll
  for (int mt = 0; mt < S::MTO; ++mt) out[mt] = z;
#pragma unroll
  for (int q = 0; q < S::Q1; ++q)
#pragma unroll
    for (int mt = 0; mt < S::MTO; ++mt) out[mt] = mfma4(F.a3(mt, q), a2[q], out[mt]);
}

// backward of one evaluation: dW into G, optionally d/d inputs (pre-tanh factor not
// applied) in din.  a1 / a2 / b0 as produced by mnet_fwd.  Wave-uniform control flow.
template <class S, int ACT, bool DROP, bool DIN, class FP, int ROWS = IMG_ROWS>
NJ_DEV void mnet_bwd(FP& Bf, GradTiles<S>& G, lfp img_d, lfp img_a,
                     const float (&dout)[S::QO], const float (&b0)[S::Q0],
                     const float (&a1)[S::Q1], const float (&a2)[S::Q1], uint32_t k1,
                     uint32_t k2, float inv_keep, float keepf, f32x4 (&din)[DIN ? S::MTI : 1],
                     int g, int c) {
  static_assert(16 * S::NT1 <= ROWS && 16 * S::NT0 <= ROWS && 16 * S::MTO <= ROWS &&
                    16 * S::MT1 <= ROWS,
                "dW staging images hold at most ROWS units");
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[S::MT1];
  Bf.begin();
  if constexpr (S::OUT == 1) {
    // one output unit: dW3[0][4q + g] += dy a2[q] per chain (the chains are summed when the block
    // flushes, GradTiles::flush) and W3^T dy is a scaling of the weight row -- no images, no MFMAs
    const float d0 = g == 0 ? dout[0] : 0.0f;
    const float dyc = edge_reduce(d0, d0);          // the chain's dy in all four lane groups
#pragma unroll
    for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = z;
#pragma unroll
    for (int q = 0; q < S::Q1; ++q) {
      G.G3[0][q / 4][q % 4] = fmaf(dyc, a2[q], G.G3[0][q / 4][q % 4]);
      if (q < S::QW) {
        const float w = Bf.r3(q);
        acc[q / 4][q % 4] = (4 * q + 3 < S::W || 4 * q + g < S::W) ? w * dyc : 0.0f;
      }
    }
  } else {
    img_write<S::QO>(img_d, dout, g, c);
    img_write<S::Q1>(img_a, a2, g, c);
    wave_lds_sync();
    dw_accumulate<S::MTO, S::NT1>(img_d, img_a, G.G3, g, c);
#pragma unroll
    for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = z;
#pragma unroll
    for (int q = 0; q < S::QO; ++q)
#pragma unroll
      for (int mt = 0; mt < S::MT1; ++mt)
        acc[mt] = mt == S::ET ? mfma1(Bf.eb3(q), dout[q], acc[mt]) : mfma4(Bf.b3(mt, q), dout[q], acc[mt]);
    if constexpr (S::ER != 0) acc[S::ET] = edge_tile<S::ER>(acc[S::ET], g);
  }
  float d2[S::QW];
  hidden_delta_g<S::MT1, S::Q1, S::QW, ACT, DROP>(acc, a2, d2, k2, inv_keep, keepf);
  wave_lds_sync();
  img_write<S::QW>(img_d, d2, g, c);
  img_write<S::Q1>(img_a, a1, g, c);
  wave_lds_sync();
  if constexpr (S::ER != 0) dw_accumulate_edge<S::G2M>(img_d, img_a, G.G2, G.GM, G.GN, 16 * g + c, g, c);
  else dw_accumulate<S::G2M, S::G2N>(img_d, img_a, G.G2, g, c);
#pragma unroll
  for (int mt = 0; mt < S::MT1; ++mt) acc[mt] = z;
#pragma unroll
  for (int q = 0; q < S::QW; ++q)
#pragma unroll
    for (int mt = 0; mt < S::MT1; ++mt)
      acc[mt] = mt == S::ET ? mfma1(Bf.eb2(q), d2[q], acc[mt]) : mfma4(Bf.b2(mt, q), d2[q], acc[mt]);
  if constexpr (S::ER != 0) acc[S::ET] = edge_tile<S::ER>(acc[S::ET], g);
  float d1[S::QW];
  hidden_delta_g<S::MT1, S::Q1, S::QW, ACT, DROP>(acc, a1, d1, k1, inv_keep, keepf);
  wave_lds_sync();
  img_write<S::QW>(img_d, d1, g, c);
  img_write<S::Q0>(img_a, b0, g, c);
  wave_lds_sync();
  dw_accumulate<S::MT1, S::NT0>(img_d, img_a, G.G1, g, c);
  if constexpr (DIN) {
#pragma unroll
    for (int mt = 0; mt < S::MTI; ++mt) din[mt] = z;
#pragma unroll
    for (int q = 0; q < S::QW; ++q)
#pragma unroll
      for (int mt = 0; mt < S::MTI; ++mt) din[mt] = mfma4(Bf.b1(mt, q), d1[q], din[mt]);
  }
  wave_lds_sync();
}

template <bool DROP> NJ_DEV void row_keep_bits(const KArgs& a, unsigned long long gid, uint32_t tkey,
                                               uint32_t net, int g, int nq, uint32_t& k1,
                                               uint32_t& k2) {
  k1 = k2 = 0;
  if constexpr (DROP) {
    uint32_t st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1),
                             tkey, net);
    k1 = keep_bits<16>(st, a.dc.thr16);
    k2 = keep_bits<16>(st, a.dc.thr16);
  }
  (void)nq;
}

// value of unit U of a per-lane full vector v[N], selected by the lane group
template <int Q, int N> NJ_DEV float by_group(const float (&v)[N], int g) {
  const float e0 = 4 * Q + 0 < N ? v[4 * Q + 0 < N ? 4 * Q + 0 : 0] : 0.0f;
  const float e1 = 4 * Q + 1 < N ? v[4 * Q + 1 < N ? 4 * Q + 1 : 0] : 0.0f;
  const float e2 = 4 * Q + 2 < N ? v[4 * Q + 2 < N ? 4 * Q + 2 : 0] : 0.0f;
  const float e3 = 4 * Q + 3 < N ? v[4 * Q + 3 < N ? 4 * Q + 3 : 0] : 0.0f;
  return g == 0 ? e0 : (g == 1 ? e1 : (g == 2 ? e2 : e3));
}

template <int NQ, int N, int Q = 0>
NJ_DEV void fill_by_group(float (&dq)[NQ], const float (&v)[N], int g) {
  if constexpr (Q < NQ) {
    dq[Q] = by_group<Q, N>(v, g);
    fill_by_group<NQ, N, Q + 1>(dq, v, g);
  }
}

// ---- E: encoder on every observation row and every start value ------------------------
template <class C> struct EncS { using type = MS<C::ENC_IN, C::H, C::W>; };
template <class C> struct DecS { using type = MS<C::H, C::DO, C::W>; };

// b0 of the encoder: [tanh(x) (D), 1]; xv = x in D-layout
template <class C, class S>
NJ_DEV void enc_input(const float* xp, float (&b0)[S::Q0], int g) {
#pragma unroll
  for (int q = 0; q < S::Q0; ++q) {
    const int u = 4 * q + g;
    const float xv = xp[u < C::D ? u : 0];
    b0[q] = u < C::D ? tanh_f(xv) : (u == C::D ? 1.0f : 0.0f);
  }
}
// encoder output + identity path (FFNN residual cases) for unit u
template <class C> NJ_DEV float enc_residual(const float* xp, int u) {
  if constexpr (C::ENC_CASE == 1) {
    return xp[u % C::D];
  } else if constexpr (C::ENC_CASE == 2) {
    constexpr int mult = C::D / C::H;
    float s = 0.0f;
#pragma unroll
    for (int cc = 0; cc < mult; ++cc) s += xp[cc * C::H + (u < C::H ? u : 0)];
    return s * (1.0f / mult);
  } else {
    return 0.0f;
  }
}

template <class C, bool DROP>
__global__ void __launch_bounds__(64) k_encode_rows_mfma(KArgs a) {
  using S = typename EncS<C>::type;
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  FwdFrags<S> F;
  F.load(a.frag_enc, lane);
  const int total = a.n_obs + a.B;
  const int n_tiles = (total + 15) / 16;
  float* const trash = a.trash + lane * C::H;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int t0 = tile * 16 + c;
    const bool valid = t0 < total;
    const int tid = valid ? t0 : 0;
    const bool is_row = tid < a.n_obs;
    const int b = is_row ? a.obs_idx[tid] : tid - a.n_obs;
    const float* xp = is_row ? a.X + (size_t)tid * C::D : a.start_X + (size_t)b * C::D;
    float b0[S::Q0], a1[S::Q1], a2[S::Q1];
    enc_input<C, S>(xp, b0, g);
    uint32_t k1, k2;
    row_keep_bits<DROP>(a, a.gid0 + b, is_row ? (uint32_t)a.k_jump[a.t_of_row[tid]] : TKEY_START,
                        NET_ENC, g, S::Q1, k1, k2);
    f32x4 out[S::MTO];
    mnet_fwd<S, C::ACT, DROP>(F, b0, a1, a2, out, k1, k2, a.dc.inv_keep, g);
    float* dstrow = valid ? (is_row ? a.h0row + (size_t)tid * C::H : a.h0start + (size_t)b * C::H)
                          : trash;
#pragma unroll
    for (int q = 0; q < S::QO; ++q) {
      const int u = 4 * q + g;
      const float v = out[q / 4][q % 4] + enc_residual<C>(xp, u < C::H ? u : 0);
      float* dst = u < C::H ? dstrow + u : trash;
      *dst = v;
    }
  }
}

// NJODE_ENC_FUSED (njode_ode2.h, ode2_item_start): the encoder evaluations the one-wave role of the
// ODE forward does NOT do itself -- the start of every item of the four-wave role's tiles (sorted
// items [0, 16 T), T = the forward's split point, on the device) and, per path, the state after its
// LAST observation (no item starts there; the row pass and the tails read it; a path without
// observations: its start value).  Tiles: [0, ceil(B / 16)) the paths, behind them the item tiles.
template <class C, bool DROP>
__global__ void __launch_bounds__(64) k_encode_rows_items(KArgs a, int n_path_tiles) {
  using S = typename EncS<C>::type;
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  FwdFrags<S> F;
  F.load(a.frag_enc, lane);
  const int T = (int)a.base_s[a.K + 2];
  const int n_tiles = n_path_tiles + (a.n_obs > 0 ? T : 0);
  float* const trash = a.trash + lane * C::H;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    bool valid, is_row;
    int b, row;
    if (tile < n_path_tiles) {
      b = tile * 16 + c;
      valid = b < a.B;
      b = valid ? b : 0;
      row = a.last_row[b];
      is_row = row >= 0;
    } else {
      const int j = (tile - n_path_tiles) * 16 + c;
      valid = j < a.n_obs;
      const int r = a.order[valid ? j : 0];
      b = a.obs_idx[r];
      row = a.item_prev[r];
      is_row = row >= 0;
    }
    const int rr = is_row ? row : 0;
    const float* xp = is_row ? a.X + (size_t)rr * C::D : a.start_X + (size_t)b * C::D;
    float b0[S::Q0], a1[S::Q1], a2[S::Q1];
    enc_input<C, S>(xp, b0, g);
    uint32_t k1, k2;
    row_keep_bits<DROP>(a, a.gid0 + b, is_row ? (uint32_t)a.k_jump[a.t_of_row[rr]] : TKEY_START, NET_ENC, g,
                        S::Q1, k1, k2);
    f32x4 out[S::MTO];
    mnet_fwd<S, C::ACT, DROP>(F, b0, a1, a2, out, k1, k2, a.dc.inv_keep, g);
    float* dstrow = valid ? (is_row ? a.h0row + (size_t)rr * C::H : a.h0start + (size_t)b * C::H) : trash;
#pragma unroll
    for (int q = 0; q < S::QO; ++q) {
      const int u = 4 * q + g;
      const float v = out[q / 4][q % 4] + enc_residual<C>(xp, u < C::H ? u : 0);
      float* dst = u < C::H ? dstrow + u : trash;
      *dst = v;
    }
  }
}

// readout input [tanh(h) (H), 1] from a row of H floats
template <class C, class S>
NJ_DEV void dec_input(const float* hp, float (&b0)[S::Q0], int g) {
#pragma unroll
  for (int q = 0; q < S::Q0; ++q) {
    const int u = 4 * q + g;
    const float hv = hp[u < C::H ? u : 0];
    b0[q] = u < C::H ? tanh_f(hv) : (u == C::H ? 1.0f : 0.0f);
  }
}
// full readout vector y[DO] of this lane's row: network output (D-layout, staged through
// an LDS image) + identity path
template <class C, class S>
NJ_DEV void dec_collect(lfp img, const f32x4 (&out)[S::MTO], const float* hp, float (&y)[C::DO],
                        int g, int c) {
  if constexpr (C::DO == 1) {   // (one output unit: mnet_fwd leaves it in EVERY lane, no image)
    float v = out[0][0];
    if constexpr (C::DEC_CASE == 1) {
      v += hp[0];
    } else if constexpr (C::DEC_CASE == 2) {
      float s = 0.0f;
#pragma unroll
      for (int cc = 0; cc < C::H; ++cc) s += hp[cc];
      v += s * (1.0f / C::H);
    }
    y[0] = v;
    return;
  }
  float o[S::QO];
#pragma unroll
  for (int q = 0; q < S::QO; ++q) o[q] = out[q / 4][q % 4];
  img_write<S::QO>(img, o, g, c);
  wave_lds_sync();
#pragma unroll
  for (int u = 0; u < C::DO; ++u) {
    float v = img[u * IMG_STRIDE + c];
    if constexpr (C::DEC_CASE == 1) {
      v += hp[u % C::H];
    } else if constexpr (C::DEC_CASE == 2) {
      constexpr int mult = C::H / C::DO;
      float s = 0.0f;
#pragma unroll
      for (int cc = 0; cc < mult; ++cc) s += hp[cc * C::DO + u];
      v += s * (1.0f / mult);
    }
    y[u] = v;
  }
  wave_lds_sync();
}

// ---- A (forward): readout before / after the jump and the loss term of each row ---------
template <class C, bool DROP>
__global__ void __launch_bounds__(64) k_jump_rows_mfma(KArgs a) {
  using S = typename DecS<C>::type;
  static_assert(C::DO <= 16, "row kernels hold the full readout vector per lane");
  __shared__ __attribute__((aligned(16))) float lds_raw[IMG_FLOATS];
  lfp img = (lfp)lds_raw;
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  FwdFrags<S> F;
  F.load(a.frag_dec, lane);
  const int n_tiles = (a.n_obs + 15) / 16;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int r0 = tile * 16 + c;
    const bool valid = r0 < a.n_obs;
    const int r = valid ? r0 : 0;
    const int b = a.obs_idx[r];
    const unsigned long long gid = a.gid0 + b;
    const uint32_t tkey = (uint32_t)a.k_jump[a.t_of_row[r]];
    const float *he = a.h_end + (size_t)r * C::H, *h0 = a.h0row + (size_t)r * C::H;
    float b0[S::Q0], a1[S::Q1], a2[S::Q1], y[C::DO], ybj[C::DO], x[C::D], mask[C::D];
    f32x4 out[S::MTO];
    uint32_t k1, k2;
    dec_input<C, S>(he, b0, g);
    row_keep_bits<DROP>(a, gid, tkey, NET_DEC_BJ, g, S::Q1, k1, k2);
    mnet_fwd<S, C::ACT, DROP>(F, b0, a1, a2, out, k1, k2, a.dc.inv_keep, g);
    dec_collect<C, S>(img, out, he, ybj, g, c);
    dec_input<C, S>(h0, b0, g);
    row_keep_bits<DROP>(a, gid, tkey, NET_DEC, g, S::Q1, k1, k2);
    mnet_fwd<S, C::ACT, DROP>(F, b0, a1, a2, out, k1, k2, a.dc.inv_keep, g);
    dec_collect<C, S>(img, out, h0, y, g, c);
    load_vec(a.X + (size_t)r * C::D, x);
#pragma unroll
    for (int i = 0; i < C::D; ++i) mask[i] = 1.0f;
    float dy[C::DO], dybj[C::DO];
    const float scale = a.inv_batch * __builtin_amdgcn_rcpf((float)a.n_obs_ot[b]);
    const float term = loss_row<C>(x, mask, y, ybj, a.weight, a.loss_easy, scale, dy, dybj);
    if (valid && g == 0) a.loss_terms[r] = term;
  }
}

// gradient w.r.t. the readout's input row from din (pre-tanh) and the identity path
template <class C, class S>
NJ_DEV void dec_input_grad(const f32x4 (&din)[S::MTI], const float (&b0)[S::Q0],
                           const float (&dyv)[C::DO], float* dst, float* trash, int g) {
#pragma unroll
  for (int q = 0; q < S::QI; ++q) {
    const int u = 4 * q + g;
    const float th = b0[q];
    float v = din[q / 4][q % 4] * (1.0f - th * th);
    if constexpr (C::DEC_CASE == 1) {
      // y[j] += h[j % H]: every output j with j % H == u
#pragma unroll
      for (int jj = 0; jj < C::DO; ++jj) v += (jj % C::H) == u ? dyv[jj] : 0.0f;
    } else if constexpr (C::DEC_CASE == 2) {
      constexpr int mult = C::H / C::DO;
      float e = 0.0f;
#pragma unroll
      for (int jj = 0; jj < C::DO; ++jj) e = (u % C::DO) == jj ? dyv[jj] : e;
      v += e * (1.0f / mult);
    }
    float* p = u < C::H ? dst + u : trash;
    *p = v;
  }
}

// ---- A (backward) ---------------------------------------------------------------------------
template <class C, bool DROP>
__global__ void __launch_bounds__(256, 2) k_jump_rows_bwd_mfma(KArgs a) {
  using S = typename DecS<C>::type;
  using NL = typename C::Dec;
  __shared__ __attribute__((aligned(16))) float lds_raw[4 * 2 * IMG_FLOATS + S::NALL * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int wave = blockIdx.x * 4 + wv, n_waves = gridDim.x * 4;
  lfp img_d = (lfp)lds_raw + wv * 2 * IMG_FLOATS, img_a = img_d + IMG_FLOATS;
  lfp fimg = (lfp)lds_raw + 4 * 2 * IMG_FLOATS;
  LdsFrags<S>::stage(fimg, a.frag_dec, threadIdx.x, 256);
  for (int i = threadIdx.x; i < 4 * 2 * IMG_FLOATS; i += 256) lds_raw[i] = 0.0f;
  __syncthreads();
  LdsFrags<S> F;
  F.init(fimg, lane);
  LdsFrags<S>& Bf = F;
  GradTiles<S> G;
  G.zero();
  float* const trash = a.trash + threadIdx.x * C::H;
  const int n_tiles = (a.n_obs + 15) / 16;
  for (int tile = wave; tile < n_tiles; tile += n_waves) {
    const int r0 = tile * 16 + c;
    const bool valid = r0 < a.n_obs;
    const int r = valid ? r0 : 0;
    const int b = a.obs_idx[r];
    const unsigned long long gid = a.gid0 + b;
    const uint32_t tkey = (uint32_t)a.k_jump[a.t_of_row[r]];
    const float *he = a.h_end + (size_t)r * C::H, *h0 = a.h0row + (size_t)r * C::H;
    float b0[S::Q0], a1[S::Q1], a2[S::Q1], y[C::DO], ybj[C::DO], x[C::D], mask[C::D];
    float dy[C::DO], dybj[C::DO], dq[S::QO];
    f32x4 out[S::MTO], din[S::MTI];
    uint32_t kb1, kb2, k1, k2;
    // y_bj, then y: both evaluations' activations stay in registers for their backward
    float b0b[S::Q0], a1b[S::Q1], a2b[S::Q1];
    dec_input<C, S>(he, b0b, g);
    row_keep_bits<DROP>(a, gid, tkey, NET_DEC_BJ, g, S::Q1, kb1, kb2);
    mnet_fwd<S, C::ACT, DROP>(F, b0b, a1b, a2b, out, kb1, kb2, a.dc.inv_keep, g);
    dec_collect<C, S>(img_d, out, he, ybj, g, c);
    dec_input<C, S>(h0, b0, g);
    row_keep_bits<DROP>(a, gid, tkey, NET_DEC, g, S::Q1, k1, k2);
    mnet_fwd<S, C::ACT, DROP>(F, b0, a1, a2, out, k1, k2, a.dc.inv_keep, g);
    dec_collect<C, S>(img_d, out, h0, y, g, c);
    load_vec(a.X + (size_t)r * C::D, x);
#pragma unroll
    for (int i = 0; i < C::D; ++i) mask[i] = 1.0f;
    const float scale = (valid ? a.inv_batch : 0.0f) * __builtin_amdgcn_rcpf((float)a.n_obs_ot[b]);
    const float term = loss_row<C>(x, mask, y, ybj, a.weight, a.loss_easy, scale, dy, dybj);
    if (a.defer_loss && valid && g == 0) a.loss_terms[r] = term;   // fused step: the loss itself
    // backward through y = readout(h0row[r])
    fill_by_group<S::QO, C::DO>(dq, dy, g);
    mnet_bwd<S, C::ACT, DROP, true>(Bf, G, img_d, img_a, dq, b0, a1, a2, k1, k2, a.dc.inv_keep,
                                    a.keep, din, g, c);
    dec_input_grad<C, S>(din, b0, dy, valid ? a.g_h0 + (size_t)r * C::H : trash, trash, g);
    // backward through y_bj = readout(h_end[r])
    fill_by_group<S::QO, C::DO>(dq, dybj, g);
    mnet_bwd<S, C::ACT, DROP, true>(Bf, G, img_d, img_a, dq, b0b, a1b, a2b, kb1, kb2,
                                    a.dc.inv_keep, a.keep, din, g, c);
    dec_input_grad<C, S>(din, b0b, dybj, valid ? a.lam_end + (size_t)r * C::H : trash, trash, g);
  }
  static_assert(3 * GradTiles<S>::NG * 256 <= 4 * 2 * IMG_FLOATS + S::NALL * 64, "tile reduction does not fit");
  if (G.reduce_block((lfp)lds_raw, wv, lane))
    G.template flush<NL>(a.slab + (size_t)blockIdx.x * C::P + C::OFF_DEC, g, c);
}

// ---- D: d loss / d encoder params -----------------------------------------------------------
// (the body as a function of the block index: njode_chain_dw.h runs it as a role of the ODE network's
// weight-gradient launch -- both only wait for the sweep)
template <class C> struct EncBwdLds {
  static constexpr int FLOATS = 4 * 2 * IMG_FLOATS + EncS<C>::type::NALL * 64;
};
template <class C, bool DROP>
NJ_DEV void encode_rows_bwd_body(const KArgs& a, float* lds_raw, int block, int n_blocks) {
  using S = typename EncS<C>::type;
  using NL = typename C::Enc;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int wave = block * 4 + wv, n_waves = n_blocks * 4;
  lfp img_d = (lfp)lds_raw + wv * 2 * IMG_FLOATS, img_a = img_d + IMG_FLOATS;
  lfp fimg = (lfp)lds_raw + 4 * 2 * IMG_FLOATS;
  LdsFrags<S>::stage(fimg, a.frag_enc, threadIdx.x, 256);
  for (int i = threadIdx.x; i < 4 * 2 * IMG_FLOATS; i += 256) lds_raw[i] = 0.0f;
  __syncthreads();
  LdsFrags<S> F;
  F.init(fimg, lane);
  LdsFrags<S>& Bf = F;
  GradTiles<S> G;
  G.zero();
  const int total = a.n_obs + a.B;
  const int n_tiles = (total + 15) / 16;
  for (int tile = wave; tile < n_tiles; tile += n_waves) {
    const int t0 = tile * 16 + c;
    const bool valid = t0 < total;
    const int tid = valid ? t0 : 0;
    const bool is_row = tid < a.n_obs;
    const int b = is_row ? a.obs_idx[tid] : tid - a.n_obs;
    const float* xp = is_row ? a.X + (size_t)tid * C::D : a.start_X + (size_t)b * C::D;
    // adjoint of this start state (see k_encode_rows_bwd)
    const int nxt = is_row ? a.item_next[tid] : a.first_row[b];
    const int nx = nxt >= 0 ? nxt : 0, rr = is_row ? tid : 0;
    float gh[S::QO];
#pragma unroll
    for (int q = 0; q < S::QO; ++q) {
      const int u = 4 * q + g, uu = u < C::H ? u : 0;
      const float l = a.lam_start[(size_t)nx * C::H + uu];
      const float p = a.g_h0[(size_t)rr * C::H + uu];
      gh[q] = (valid && u < C::H) ? ((nxt >= 0 ? l : 0.0f) + (is_row ? p : 0.0f)) : 0.0f;
    }
    float b0[S::Q0], a1[S::Q1], a2[S::Q1];
    enc_input<C, S>(xp, b0, g);
    uint32_t k1, k2;
    row_keep_bits<DROP>(a, a.gid0 + b, is_row ? (uint32_t)a.k_jump[a.t_of_row[tid]] : TKEY_START,
                        NET_ENC, g, S::Q1, k1, k2);
    f32x4 out[S::MTO], din[1];
    mnet_fwd<S, C::ACT, DROP>(F, b0, a1, a2, out, k1, k2, a.dc.inv_keep, g);
    mnet_bwd<S, C::ACT, DROP, false>(Bf, G, img_d, img_a, gh, b0, a1, a2, k1, k2, a.dc.inv_keep,
                                     a.keep, din, g, c);
  }
  static_assert(3 * GradTiles<S>::NG * 256 <= 4 * 2 * IMG_FLOATS + S::NALL * 64, "tile reduction does not fit");
  if (G.reduce_block((lfp)lds_raw, wv, lane))
    G.template flush<NL>(a.slab + (size_t)block * C::P + C::OFF_ENC, g, c);
}
template <class C, bool DROP>
__global__ void __launch_bounds__(256, 2) k_encode_rows_bwd_mfma(KArgs a) {
  __shared__ __attribute__((aligned(16))) float lds_raw[EncBwdLds<C>::FLOATS];
  encode_rows_bwd_body<C, DROP>(a, lds_raw, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace njode
