// Cooperative Poseidon2: ONE permutation spread over 16 adjacent lanes (one DPP
// "row"), state element i in lane i of the row, four permutations per wave.
//
// The lane-per-state form (poseidon2.cuh) is the throughput form: 64 permutations
// per wave, about 6 900 VALU instructions each, a dependent chain about 15 us long.
// Tree tops, FRI tails and the Fiat-Shamir sponge are chains of a handful of
// DEPENDENT permutations with almost no parallelism, so their cost is that
// latency.  Here a permutation is about 900 instructions: every lane raises its
// own element to the 7th power and the linear layers are cross-lane DPP adds
// (quad_perm inside a 4-chunk, row_ror across chunks), about 5x shorter.
// Same function, same constants, bit-identical results.
#pragma once
#include "poseidon2.cuh"

namespace zksp {

template <int CTRL>
__device__ __forceinline__ Fp dpp(Fp v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return Fp::raw((uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.v, CTRL, 0xf, 0xf, false));
#else
  return v;  // host pass only parses this header
#endif
}
constexpr int kQuadRot1 = 0x39;   // quad_perm:[1,2,3,0]  lane j <- lane (j+1)%4
constexpr int kQuadSwap1 = 0xB1;  // quad_perm:[1,0,3,2]
constexpr int kQuadSwap2 = 0x4E;  // quad_perm:[2,3,0,1]
constexpr int kRowRor1 = 0x121, kRowRor2 = 0x122, kRowRor4 = 0x124, kRowRor8 = 0x128;

// Per-lane constants of one row member (element index e = lane & 15).
struct CoopConsts {
  Fp ext[8];   // external round constants of element e
  Fp diag;     // internal diagonal entry of element e
  bool is0;    // e == 0: the element the internal S-box acts on
};

__device__ __forceinline__ CoopConsts coop_load_consts(const P2Consts* __restrict__ k, int e) {
  CoopConsts c;
#pragma unroll
  for (int r = 0; r < 8; ++r) c.ext[r] = Fp::raw(k->ext[r][e]);
  c.diag = Fp::raw(k->diag[e]);
  c.is0 = (e == 0);
  return c;
}

__device__ __forceinline__ Fp coop_sbox(Fp x) {
  // same lazy-reduction bounds as p2_sbox_layer
  uint32_t a = x.v, x2, x3, x4, x7;
  fp_mul_batch_raw<1>(&x2, &a, &a);
  fp_mul_batch_raw<1>(&x3, &x2, &a);
  fp_mul_batch_raw<1>(&x4, &x2, &x2);
  x4 = fp_correct(x4);
  fp_mul_batch_raw<1>(&x7, &x3, &x4);
  return Fp::raw(fp_correct(x7));
}

// circ(2*M4, M4, M4, M4): y_j = t + x_j + 2 x_{j+1} inside the quad, then add the
// column sums over the four quads
__device__ __forceinline__ Fp coop_external_linear(Fp x) {
  Fp a = x + dpp<kQuadSwap1>(x);
  Fp t = a + dpp<kQuadSwap2>(a);
  Fp y = t + x + dpp<kQuadRot1>(x).dbl();
  Fp u = y + dpp<kRowRor8>(y);
  Fp v = u + dpp<kRowRor4>(u);
  return y + v;
}

__device__ __forceinline__ Fp coop_row_sum(Fp x) {
  Fp s = x + dpp<kRowRor8>(x);
  s = s + dpp<kRowRor4>(s);
  s = s + dpp<kRowRor2>(s);
  return s + dpp<kRowRor1>(s);
}

// x: this lane's state element.  All 16 lanes of the row must be active.
__device__ __forceinline__ Fp p2_permute_coop(Fp x, const CoopConsts& c, const P2Consts* __restrict__ k) {
  x = coop_external_linear(x);
#pragma unroll
  for (int r = 0; r < 4; ++r) x = coop_external_linear(coop_sbox(x + c.ext[r]));
#pragma unroll 1
  for (int r = 0; r < 13; ++r) {
    Fp sb = coop_sbox(x + Fp::raw(k->internal[r]));
    x = c.is0 ? sb : x;
    x = x * c.diag + coop_row_sum(x);
  }
#pragma unroll
  for (int r = 4; r < 8; ++r) x = coop_external_linear(coop_sbox(x + c.ext[r]));
  return x;
}

}  // namespace zksp
