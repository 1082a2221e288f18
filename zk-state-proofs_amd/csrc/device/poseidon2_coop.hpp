// Cooperative Poseidon2: ONE permutation spread over 16 adjacent lanes (one DPP
// "row"), state element i in lane i of the row, four permutations per wave.
//
// The lane-per-state form (poseidon2.hpp) is the throughput form: 64 permutations
// per wave, about 5 000 VALU instructions each.  Tree tops, FRI tails, the
// Fiat-Shamir sponge and a single proof's leaf hashing are chains of DEPENDENT
// permutations with almost no parallelism, so their cost is the length of the
// dependency chain.  Here every lane raises its own element to the 7th power and
// the linear layers are cross-lane DPP adds (quad_perm inside a 4-chunk, row_ror
// across chunks).
//
// Arithmetic is the signed lazy layer of field.hpp, which roughly halves the chain
// (about 450 dependent instructions instead of 870 with canonical residues):
//  * a product is mad, mul_lo, mad with no correction;
//  * a word x is split as x = xh * 2^16 + xl (xh signed, xl in [0, 2^16)) before a linear
//    layer; the DPP adds run on the two halves independently and cannot overflow (the
//    weights of a row sum to 35, so |yh| < 2^21 and yl < 2^22);
//  * one reduction recombines them: T = yh * (2^16 c) + yl * c + rc R^2 == (y + rc~) R,
//    c = 2^32 mod p, followed by the centred Montgomery reduction; |result| < 0.51p;
//  * internal rounds: T = x * d~ + (sum_h * (2^16 c) + sum_l * c) [+ rc R^2 on element 0].
// Same function, same constants, bit-identical results after canonicalisation.
#pragma once
#include "poseidon2.hpp"

namespace zksp {

template <int CTRL>
__device__ __forceinline__ int32_t dpp_i(int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
#else
  return v;  // host pass only parses this header
#endif
}
constexpr int kQuadRot1 = 0x39;   // quad_perm:[1,2,3,0]  lane j <- lane (j+1)%4
constexpr int kQuadSwap1 = 0xB1;  // quad_perm:[1,0,3,2]
constexpr int kQuadSwap2 = 0x4E;  // quad_perm:[2,3,0,1]
constexpr int kRowRor1 = 0x121, kRowRor2 = 0x122, kRowRor4 = 0x124, kRowRor8 = 0x128;

// (2^16 * c) mod p, centred: the weight of the high half in a recombining reduction
constexpr int32_t kC16Centred = fps_centre_const((uint32_t)((((uint64_t)kRModP) << 16) % kP));

// Per-lane constants of one row member (element index e = lane & 15).
struct CoopConsts {
  int64_t lin_add[9];  // P2Consts::lin_add[l][e]
  int64_t int_last;    // P2Consts::int_last[e]
  int32_t sdiag;       // P2Consts::sdiag[e]
  bool is0;            // e == 0: the element the internal S-box acts on
};

__device__ __forceinline__ CoopConsts coop_load_consts(const P2Consts* __restrict__ k, int e) {
  CoopConsts c;
#pragma unroll
  for (int l = 0; l < 9; ++l) c.lin_add[l] = k->lin_add[l][e];
  c.int_last = k->int_last[e];
  c.sdiag = k->sdiag[e];
  c.is0 = (e == 0);
  return c;
}

// halves of a signed word and the reduction that recombines a pair of half-sums
__device__ __forceinline__ void coop_split(int32_t x, int32_t& h, int32_t& l) {
  l = x & 0xffff;
  h = x >> 16;
}
__device__ __forceinline__ int64_t coop_recombine(int32_t h, int32_t l, int64_t add) {
  return (int64_t)h * (int64_t)kC16Centred + ((int64_t)((uint64_t)(uint32_t)l * kRModP) + add);
}

// circ(2*M4, M4, M4, M4): y_j = t + x_j + 2 x_{j+1} inside the quad, then add the
// column sums over the four quads; on one half of the words
__device__ __forceinline__ int32_t coop_external_half(int32_t x) {
  const int32_t a = x + dpp_i<kQuadSwap1>(x);
  const int32_t t = a + dpp_i<kQuadSwap2>(a);
  const int32_t y = t + x + 2 * dpp_i<kQuadRot1>(x);
  const int32_t u = y + dpp_i<kRowRor8>(y);
  const int32_t v = u + dpp_i<kRowRor4>(u);
  return y + v;
}
__device__ __forceinline__ int32_t coop_external_linear(int32_t x, int64_t add) {
  int32_t h, l;
  coop_split(x, h, l);
  return fps_redc(coop_recombine(coop_external_half(h), coop_external_half(l), add));
}

__device__ __forceinline__ int32_t coop_row_sum_half(int32_t x) {
  int32_t s = x + dpp_i<kRowRor8>(x);
  s = s + dpp_i<kRowRor4>(s);
  s = s + dpp_i<kRowRor2>(s);
  return s + dpp_i<kRowRor1>(s);
}

// x: this lane's state element, a signed word with |x| < 1.034p (canonical qualifies); same on
// exit (|x| < 0.51p after the last layer).  All 16 lanes of the row must be active.
__device__ __forceinline__ int32_t p2_permute_coop_signed(int32_t x, const CoopConsts& c,
                                                          const P2Consts* __restrict__ k) {
  x = coop_external_linear(x, c.lin_add[0]);
#pragma unroll
  for (int r = 0; r < 4; ++r) x = coop_external_linear(p2s_sbox(x), c.lin_add[r + 1]);
#pragma unroll 1
  for (int r = 0; r < 13; ++r) {
    const int32_t sb = p2s_sbox(x);
    x = c.is0 ? sb : x;
    int32_t h, l;
    coop_split(x, h, l);
    // the next constant: element 0 only after rounds 0..11, every element after round 12
    const int64_t add = r < 12 ? (c.is0 ? k->int_add[r] : 0) : c.int_last;
    x = fps_redc((int64_t)x * (int64_t)c.sdiag + coop_recombine(coop_row_sum_half(h), coop_row_sum_half(l), add));
  }
#pragma unroll
  for (int r = 4; r < 8; ++r) x = coop_external_linear(p2s_sbox(x), c.lin_add[r + 1]);
  return x;
}

__device__ __forceinline__ Fp p2_permute_coop(Fp x, const CoopConsts& c, const P2Consts* __restrict__ k) {
  return Fp::raw(fps_canon(p2_permute_coop_signed((int32_t)x.v, c, k)));
}

}  // namespace zksp
