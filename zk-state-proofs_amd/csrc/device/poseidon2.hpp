// Poseidon2 over BabyBear, width 16, x^7, 4+13+4 rounds: permutation, sponge and
// 2-to-1 compression.  Replaces p3-poseidon2 / p3-symmetric 0.1.4-succinct
// (reference Cargo.lock:5353, :5367).  Parameterisation and round constants are
// this repository's own (DESIGN.md "Poseidon2 instance"); constants arrive in
// Montgomery form through a P2Consts table built on the host.
#pragma once
#include "field.hpp"

namespace zksp {

struct P2Consts {
  uint32_t ext[8][16];  // external round constants (4 initial, 4 terminal)
  uint32_t internal[13];
  uint32_t diag[16];    // internal diagonal [-2, 1, 2, 4, ..., 8192, 32768]
  // Derived tables of the signed lazy permutation below (built by build_p2_consts).
  // Every round constant is added inside the reduction that closes the linear layer
  // before it, as rc * R^2 mod p (the reduction divides by R = 2^32):
  int32_t sdiag[16];       // diag centred into (-p/2, p/2]
  int64_t lin_add[9][16];  // layer 0 = initial linear layer, layer r+1 closes external round r
  uint32_t lin_rc[9][16];  // the same constants as plain Montgomery-form words (joined to a wide sum)
  int64_t int_add[13];     // element 0 after internal round r (r = 0..11)
  int64_t int_last[16];    // all elements after internal round 12
};

// ---------------------------------------------------------------------------
// Signed lazy permutation.
//
// The permutation is bound by vector-ALU issue, and with canonical residues half of
// its instructions are modular additions (add, subtract p, min) because p = 0.94 * 2^31
// leaves no headroom in a 32-bit lane.  Here a state element is instead ANY int32 t
// congruent to the value (Montgomery form), |t| < 2^31 = 1.0667p, and:
//
//  * products use a centred Montgomery reduction: m = lo(T) * (-p^-1) taken as a SIGNED
//    word, t = (T + m p) / 2^32, so |t| <= |T| / 2^32 + p/2.  For |a|, |b| < 1.0667p this
//    gives |t| < 1.034p with |T + m p| < 2^63: closed under multiplication with no
//    conditional subtraction at all (3 instructions: mad_i64_i32, mul_lo, mad_i64_i32);
//  * the external layers accumulate in 64 bits (|y| < 36.8p, one instruction per term) and each
//    output is brought back below 0.55p by an estimate of y / p from its top bits and one
//    multiplication (fps_reduce_small); the next round's constant is added to the reduced word;
//  * an internal round is sum = sum of the 16 words (64-bit, = hi * 2^32 + lo), SR = lo * c + hi * K
//    == sum * R (mod p) with c = 2^32 mod p and K = c^2 mod p, and out_i = redc(x_i * d_i~ + SR)
//    with d_i~ centred: |T| < 0.525 p^2 + 2^60.1, |t| < 0.88p.
//
// Outputs are canonicalised only where a canonical word is needed (digests); a sponge
// keeps its state signed between permutations.  Results are bit-identical to the
// canonical implementation after canonicalisation (tests/test_gpu_kernels.py and the
// host verifier, which runs this same code).
// ---------------------------------------------------------------------------
// the arithmetic itself is field.hpp's signed lazy layer
constexpr int32_t p2s_centre(uint32_t v) { return fps_centre_const(v); }
ZKSP_HD int32_t p2s_sbox(int32_t x) {
  const int32_t x2 = fps_mul(x, x), x3 = fps_mul(x2, x), x4 = fps_mul(x2, x2);
  return fps_mul(x3, x4);
}

// circ(2*M4, M4, M4, M4), M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]], then + next constants.
// The sums run in 64 bits (|y| < 36.8p < 2^36.2; a 64-bit addition issues at the rate of any other VOP3
// instruction on gfx950, see profiles/r03_opcode_rates.txt), nine additions per 4-block; fps_reduce_small brings
// each back below 0.55p with one multiply, and rc[i], the next round's constant (a canonical Montgomery-form
// word), joins the reduced word centred, with a 32-bit addition: |out| < 1.05p, inside what the S-box and
// the internal round accept (|x| < 2^31 = 1.0667p; x * x stays below the 1.209 p^2 fps_redc needs).
ZKSP_HD void p2s_external_linear(int32_t* s, const uint32_t* __restrict__ rc) {
  int64_t y[16];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t x0 = s[4 * c], x1 = s[4 * c + 1], x2 = s[4 * c + 2], x3 = s[4 * c + 3];
    const int64_t t01 = x0 + x1, t23 = x2 + x3, sum = t01 + t23, a = sum + x1, b = sum + x3;
    y[4 * c] = a + t01;         // 2 x0 + 3 x1 + x2 + x3
    y[4 * c + 1] = a + 2 * x2;  // x0 + 2 x1 + 3 x2 + x3
    y[4 * c + 2] = b + t23;     // x0 + x1 + 2 x2 + 3 x3
    y[4 * c + 3] = b + 2 * x0;  // 3 x0 + x1 + x2 + 2 x3
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t col = (y[j] + y[4 + j]) + (y[8 + j] + y[12 + j]);
#pragma unroll
    for (int c = 0; c < 4; ++c) s[4 * c + j] = fps_reduce_small(y[4 * c + j] + col) + fps_centre(rc[4 * c + j]);
  }
}

// x0 -> x0^7 (its constant was added by the previous reduction), then y_i = d_i x_i + sum(x).
// FULL: every output gets an addend (the layer before the terminal external rounds);
// otherwise only element 0 does.
template <bool FULL>
ZKSP_HD void p2s_internal_round(int32_t* s, const P2Consts* __restrict__ k, const int64_t* __restrict__ add) {
  s[0] = p2s_sbox(s[0]);
  int64_t sum_a = (int64_t)s[0] + s[1], sum_b = (int64_t)s[2] + s[3];
#pragma unroll
  for (int i = 4; i < 16; i += 2) {
    sum_a += s[i];
    sum_b += s[i + 1];
  }
  const int64_t sum = sum_a + sum_b;
  const uint32_t lo = (uint32_t)sum;
  const int32_t hi = (int32_t)(sum >> 32);
  const int64_t sr = (int64_t)hi * (int64_t)kR2Centred + (int64_t)((uint64_t)lo * kRModP);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int64_t t = (int64_t)s[i] * (int64_t)k->sdiag[i] + sr;
    if (FULL || i == 0) t += add[FULL ? i : 0];
    s[i] = fps_redc(t);
  }
}

// state: signed words, |s_i| < 1.05p (canonical residues qualify); |s_i| < 0.55p on exit
ZKSP_HD void p2_permute_signed(int32_t* s, const P2Consts* __restrict__ k) {
  p2s_external_linear(s, k->lin_rc[0]);
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = p2s_sbox(s[i]);
    p2s_external_linear(s, k->lin_rc[r + 1]);
  }
#pragma unroll 1
  for (int r = 0; r < 12; ++r) p2s_internal_round<false>(s, k, &k->int_add[r]);
  p2s_internal_round<true>(s, k, k->int_last);
#pragma unroll 1
  for (int r = 4; r < 8; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = p2s_sbox(s[i]);
    p2s_external_linear(s, k->lin_rc[r + 1]);
  }
}

ZKSP_HD void p2_permute(Fp* s, const P2Consts* __restrict__ k) {
  int32_t t[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = (int32_t)s[i].v;
  p2_permute_signed(t, k);
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = Fp::raw(fps_canon(t[i]));
}

}  // namespace zksp
