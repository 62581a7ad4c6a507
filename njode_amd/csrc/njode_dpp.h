// njode_dpp.h -- a matrix-vector product of ONE chain without LDS: the layer's input vector lives one
// unit per lane, the lane's weight row in registers, and the broadcast of input unit k to every lane
// is the DPP modifier of the fma itself (gfx90a+: row_newbcast:n = "lane n of my row of 16").
//
// Unit <-> lane: lane 16 g + c holds unit 4 c + g (c < 16, g < 4), i.e. row g of the wave holds the
// units = g (mod 4).  dpp_replicate() turns the vector into four registers R[g'] = "row g' of the
// vector, in every row" (three v_permlane swaps), after which
//     v_fmac_f32_dpp acc, R[g'], w row_newbcast:n        acc += unit(4 n + g') * w
// is the k-th term of the lane's dot product for every lane at once: K fma instructions for K inputs,
// in k order, no LDS round trip (~130 cycles per layer for one wave: write, wait, 13 ds_read_b128
// whose latency the compiler exposes pair by pair -- tools/ubench/chain_ubench.hip), no s_waitcnt.
//
// The fma chain is inline assembly (the compiler does not fold row_newbcast moves into fma).
// Hazards the assembler does not see inside it: a DPP operand must not have been written by the two
// preceding VALU instructions -- dpp_replicate() ends with an s_nop 1, and nothing in a block writes
// R.  Blocks of up to four quads per asm statement (the compiler puts an s_nop between statements).
#pragma once
#include "njode_device.h"

namespace njode {

typedef unsigned dpp_u32x2 __attribute__((ext_vector_type(2)));

// lane -> unit of the DPP layout
NJ_DEV int dpp_unit(int lane) { return 4 * (lane & 15) + (lane >> 4); }

// own value of the vector (unit dpp_unit(lane)) -> R[g] = units = g (mod 4), replicated in every row
NJ_DEV void dpp_replicate(float own, float (&R)[4]) {
  const unsigned a = __float_as_uint(own);
  const dpp_u32x2 s = __builtin_amdgcn_permlane32_swap(a, a, false, false);   // [r0 r1 r0 r1], [r2 r3 r2 r3]
  const dpp_u32x2 p = __builtin_amdgcn_permlane16_swap(s[0], s[0], false, false);
  const dpp_u32x2 q = __builtin_amdgcn_permlane16_swap(s[1], s[1], false, false);
  R[0] = __uint_as_float(p[0]);
  R[1] = __uint_as_float(p[1]);
  R[2] = __uint_as_float(q[0]);
  R[3] = __uint_as_float(q[1]);
  asm volatile("s_nop 1" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]));
}

#define NJ_DPP_Q(w0, w1, w2, w3, n)                                                  \
  "v_fmac_f32_dpp %0, %1, %" #w0 " row_newbcast:%" #n " row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %0, %2, %" #w1 " row_newbcast:%" #n " row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %0, %3, %" #w2 " row_newbcast:%" #n " row_mask:0xf bank_mask:0xf\n\t" \
  "v_fmac_f32_dpp %0, %4, %" #w3 " row_newbcast:%" #n " row_mask:0xf bank_mask:0xf\n\t"

// NQ full quads starting at quad Q0: units 4 Q0 .. 4 (Q0 + NQ) - 1, weights w[0 .. 4 NQ)
template <int Q0, int NQ> NJ_DEV void dpp_block(float& acc, const float (&R)[4], const float* w) {
  static_assert(NQ >= 1 && NQ <= 4 && Q0 + NQ <= 16, "block of one to four quads");
  if constexpr (NQ == 4)
    asm(NJ_DPP_Q(5, 6, 7, 8, 21) NJ_DPP_Q(9, 10, 11, 12, 22) NJ_DPP_Q(13, 14, 15, 16, 23) NJ_DPP_Q(17, 18, 19, 20, 24)
        : "+v"(acc)
        : "v"(R[0]), "v"(R[1]), "v"(R[2]), "v"(R[3]), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]),
          "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]),
          "v"(w[14]), "v"(w[15]), "n"(Q0), "n"(Q0 + 1), "n"(Q0 + 2), "n"(Q0 + 3));
  else if constexpr (NQ == 3)
    asm(NJ_DPP_Q(5, 6, 7, 8, 17) NJ_DPP_Q(9, 10, 11, 12, 18) NJ_DPP_Q(13, 14, 15, 16, 19)
        : "+v"(acc)
        : "v"(R[0]), "v"(R[1]), "v"(R[2]), "v"(R[3]), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]),
          "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "n"(Q0), "n"(Q0 + 1),
          "n"(Q0 + 2));
  else if constexpr (NQ == 2)
    asm(NJ_DPP_Q(5, 6, 7, 8, 13) NJ_DPP_Q(9, 10, 11, 12, 14)
        : "+v"(acc)
        : "v"(R[0]), "v"(R[1]), "v"(R[2]), "v"(R[3]), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]),
          "v"(w[5]), "v"(w[6]), "v"(w[7]), "n"(Q0), "n"(Q0 + 1));
  else
    asm(NJ_DPP_Q(5, 6, 7, 8, 9)
        : "+v"(acc)
        : "v"(R[0]), "v"(R[1]), "v"(R[2]), "v"(R[3]), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "n"(Q0));
}
#undef NJ_DPP_Q

// the last, partial quad: CNT = 1 .. 3 units
template <int Q, int CNT> NJ_DEV void dpp_tail(float& acc, const float (&R)[4], const float* w) {
  static_assert(CNT >= 1 && CNT <= 3 && Q < 16, "partial quad");
  if constexpr (CNT == 3)
    asm("v_fmac_f32_dpp %0, %1, %4 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %3, %6 row_newbcast:%7 row_mask:0xf bank_mask:0xf"
        : "+v"(acc) : "v"(R[0]), "v"(R[1]), "v"(R[2]), "v"(w[0]), "v"(w[1]), "v"(w[2]), "n"(Q));
  else if constexpr (CNT == 2)
    asm("v_fmac_f32_dpp %0, %1, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "+v"(acc) : "v"(R[0]), "v"(R[1]), "v"(w[0]), "v"(w[1]), "n"(Q));
  else
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc) : "v"(R[0]), "v"(w[0]), "n"(Q));
}

// init + sum_{k < K} unit(k) w[k]; R from dpp_replicate.  TWO accumulators: the blocks of four quads
// alternate between them (units 0-15, 32-47 | 16-31, 48-63), so a rounding error made early is carried
// through half as many additions as in one chain of K -- the summation of the matrix-core kernels
// (njode_mfma_lock4.h: q4_dot), whose gradients sit 0.4x the reference's own distance from float64
// where one chain of 50 sat 2.4x (profiles/r06_f64_truth.txt).  Same instruction count + one add.
template <int K, int Q = 0> NJ_DEV void dpp_dot_from(float& acc0, float& acc1, const float (&R)[4], const float (&w)[K]) {
  static_assert(K <= 64, "one unit per lane");
  constexpr int FULL = K / 4 - Q;   // full quads left
  if constexpr (FULL >= 4) {
    dpp_block<Q, 4>(acc0, R, w + 4 * Q);
    dpp_dot_from<K, Q + 4>(acc1, acc0, R, w);   // (the roles swap: the next block goes to the other one)
  } else if constexpr (FULL >= 1) {
    dpp_block<Q, FULL>(acc0, R, w + 4 * Q);
    dpp_dot_from<K, Q + FULL>(acc0, acc1, R, w);   // (the tail stays with this block's accumulator)
  } else if constexpr (K % 4 != 0 && Q == K / 4) {
    dpp_tail<Q, K % 4>(acc0, R, w + 4 * Q);
  }
}
template <int K> NJ_DEV float dpp_dot(float init, const float (&R)[4], const float (&w)[K]) {
  float acc0 = init, acc1 = 0.0f;
  dpp_dot_from<K, 0>(acc0, acc1, R, w);
  return K > 16 ? acc0 + acc1 : acc0;
}

// ---- one register, no replication -----------------------------------------------------------------
// acc + sum_{n < K} w[n] * v[lane n of my row of 16]: the vector v is either the SAME in every row (a
// small vector kept row-replicated: unit n in lane n of each row -- the H <= 16 state of the demo
// models) or the row's own slice of a unit-layout vector (a K-split: row g sums the units = g mod 4,
// dpp_rows_sum() adds the four partial sums).
#define NJ_DPP_R(w, n) "v_fmac_f32_dpp %0, %1, %" #w " row_newbcast:%" #n " row_mask:0xf bank_mask:0xf\n\t"
template <int N0, int CNT> NJ_DEV void dpp_row_block(float& acc, float v, const float* w) {
  static_assert(CNT >= 1 && CNT <= 8 && N0 + CNT <= 16, "block of one to eight lanes");
  if constexpr (CNT == 8)
    asm(NJ_DPP_R(2, 10) NJ_DPP_R(3, 11) NJ_DPP_R(4, 12) NJ_DPP_R(5, 13) NJ_DPP_R(6, 14) NJ_DPP_R(7, 15) NJ_DPP_R(8, 16) NJ_DPP_R(9, 17)
        : "+v"(acc)
        : "v"(v), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "n"(N0),
          "n"(N0 + 1), "n"(N0 + 2), "n"(N0 + 3), "n"(N0 + 4), "n"(N0 + 5), "n"(N0 + 6), "n"(N0 + 7));
  else if constexpr (CNT >= 4) {
    asm(NJ_DPP_R(2, 6) NJ_DPP_R(3, 7) NJ_DPP_R(4, 8) NJ_DPP_R(5, 9)
        : "+v"(acc)
        : "v"(v), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "n"(N0), "n"(N0 + 1), "n"(N0 + 2), "n"(N0 + 3));
    if constexpr (CNT > 4) dpp_row_block<N0 + 4, CNT - 4>(acc, v, w + 4);
  } else if constexpr (CNT == 3)
    asm(NJ_DPP_R(2, 5) NJ_DPP_R(3, 6) NJ_DPP_R(4, 7)
        : "+v"(acc) : "v"(v), "v"(w[0]), "v"(w[1]), "v"(w[2]), "n"(N0), "n"(N0 + 1), "n"(N0 + 2));
  else if constexpr (CNT == 2)
    asm(NJ_DPP_R(2, 4) NJ_DPP_R(3, 5) : "+v"(acc) : "v"(v), "v"(w[0]), "v"(w[1]), "n"(N0), "n"(N0 + 1));
  else
    asm(NJ_DPP_R(2, 3) : "+v"(acc) : "v"(v), "v"(w[0]), "n"(N0));
}
#undef NJ_DPP_R
// (v must not have been written by the two preceding VALU instructions: dpp_settle)
NJ_DEV void dpp_settle(float& v) { asm volatile("s_nop 1" : "+v"(v)); }
template <int K> NJ_DEV float dpp_row_dot(float init, float v, const float (&w)[K]) {
  static_assert(K >= 1 && K <= 16, "one row of 16 lanes");
  float acc = init;
  dpp_row_block<0, (K > 8 ? 8 : K)>(acc, v, w);
  if constexpr (K > 8) dpp_row_block<8, K - 8>(acc, v, w + 8);
  return acc;
}
// sum of a value over the four rows (lanes c, 16 + c, 32 + c, 48 + c), the same in all four:
// (r0 + r2) + (r1 + r3), a fixed order
NJ_DEV float dpp_rows_sum(float v) {
  const unsigned a = __float_as_uint(v);
  const dpp_u32x2 s = __builtin_amdgcn_permlane32_swap(a, a, false, false);   // [r0 r1 r0 r1], [r2 r3 r2 r3]
  const float t = __uint_as_float(s[0]) + __uint_as_float(s[1]);               // [r0+r2, r1+r3, r0+r2, r1+r3]
  const unsigned b = __float_as_uint(t);
  const dpp_u32x2 p = __builtin_amdgcn_permlane16_swap(b, b, false, false);   // [t0 t0 t0 t0], [t1 t1 t1 t1]
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}

}  // namespace njode
