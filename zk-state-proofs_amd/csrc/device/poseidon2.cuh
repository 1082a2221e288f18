// Poseidon2 over BabyBear, width 16, x^7, 4+13+4 rounds: permutation, sponge and
// 2-to-1 compression.  Replaces p3-poseidon2 / p3-symmetric 0.1.4-succinct
// (reference Cargo.lock:5353, :5367).  Parameterisation and round constants are
// this repository's own (DESIGN.md "Poseidon2 instance"); constants arrive in
// Montgomery form through a P2Consts table built on the host.
#pragma once
#include "field.cuh"

namespace zksp {

struct P2Consts {
  uint32_t ext[8][16];  // external round constants (4 initial, 4 terminal)
  uint32_t internal[13];
  uint32_t diag[16];    // internal diagonal [-2, 1, 2, 4, ..., 8192, 32768]
};

ZKSP_HD Fp p2_sbox(Fp x) {
  Fp x2 = x.sqr(), x3 = x2 * x, x4 = x2.sqr();
  return x3 * x4;
}

// circ(2*M4, M4, M4, M4) with M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]
ZKSP_HD void p2_external_linear(Fp* s) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    Fp a = s[4 * c], b = s[4 * c + 1], cc = s[4 * c + 2], d = s[4 * c + 3];
    Fp t = a + b + cc + d;
    // row i of M4 . v = t + v_i + 2 v_{i+1}
    s[4 * c] = t + a + b.dbl();
    s[4 * c + 1] = t + b + cc.dbl();
    s[4 * c + 2] = t + cc + d.dbl();
    s[4 * c + 3] = t + d + a.dbl();
  }
  Fp sums[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) sums[j] = s[j] + s[4 + j] + s[8 + j] + s[12 + j];
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = s[i] + sums[i & 3];
}

ZKSP_HD void p2_internal_linear(Fp* s, const P2Consts* __restrict__ k) {
  Fp sum = s[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) sum = sum + s[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = s[i] * Fp::raw(k->diag[i]) + sum;
}

ZKSP_HD void p2_permute(Fp* s, const P2Consts* __restrict__ k) {
  p2_external_linear(s);
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = p2_sbox(s[i] + Fp::raw(k->ext[r][i]));
    p2_external_linear(s);
  }
#pragma unroll 1
  for (int r = 0; r < 13; ++r) {
    s[0] = p2_sbox(s[0] + Fp::raw(k->internal[r]));
    p2_internal_linear(s, k);
  }
#pragma unroll 1
  for (int r = 4; r < 8; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = p2_sbox(s[i] + Fp::raw(k->ext[r][i]));
    p2_external_linear(s);
  }
}

}  // namespace zksp
