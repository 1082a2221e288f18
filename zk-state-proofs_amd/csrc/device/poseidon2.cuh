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

#if defined(__HIP_DEVICE_COMPILE__)
#define ZKSP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define ZKSP_SCHED_FENCE() ((void)0)
#endif

// Montgomery products of N independent pairs, written stage by stage with the
// instruction order pinned (hipcc otherwise emits each product as one serial
// mad -> mul_lo -> mad chain through a shared temporary).
//
// "Raw" products skip the final conditional subtraction: for operands a, b the
// result is < a*b/2^32 + p.  With A = p/2^32 = 0.46875: reduced x reduced gives
// < 1.469p, and the S-box below keeps every intermediate inside the two
// conditions that matter: a*b + m*p < 2^64 (a*b < 1.1333 * 2^32 * p) and the
// result < 2^32.
template <int N>
ZKSP_HD void fp_mul_batch_raw(uint32_t* out, const uint32_t* a, const uint32_t* b) {
  uint64_t t[N];
  uint32_t m[N];
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = (uint64_t)a[i] * b[i];
  ZKSP_SCHED_FENCE();
#pragma unroll
  for (int i = 0; i < N; ++i) m[i] = (uint32_t)t[i] * kMontyNegMu;
  ZKSP_SCHED_FENCE();
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = t[i] + (uint64_t)m[i] * kP;
  ZKSP_SCHED_FENCE();
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = (uint32_t)(t[i] >> 32);
}

// v < 2p  ->  v mod p
ZKSP_HD uint32_t fp_correct(uint32_t v) {
  uint32_t w = v - kP;
  return v < w ? v : w;
}

template <int N>
ZKSP_HD void fp_mul_batch(Fp* out, const Fp* a, const Fp* b) {
  uint32_t r[N], av[N], bv[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { av[i] = a[i].v; bv[i] = b[i].v; }
  fp_mul_batch_raw<N>(r, av, bv);
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = Fp::raw(fp_correct(r[i]));
  ZKSP_SCHED_FENCE();
}

ZKSP_HD Fp p2_sbox(Fp x) {
  Fp x2 = x.sqr(), x3 = x2 * x, x4 = x2.sqr();
  return x3 * x4;
}

// x -> (x + rc)^7 on N lanes with lazy reduction:
//   x  < p            (reduced sum)
//   x2 = x*x   raw  < 1.469p
//   x3 = x2*x  raw  < 1.689p      (1.469 p^2 < 1.1333 * 2^32 p)
//   x4 = x2*x2 raw  < 2.012p      (2.158 p^2 = 1.0116 * 2^32 p < 1.1333 * 2^32 p; 2.012p < 2^32)
//   x4 corrected once < 1.012p
//   x7 = x3*x4 raw  < 1.801p      (1.709 p^2 < 1.1333 * 2^32 p), corrected once -> < p
template <int N>
ZKSP_HD void p2_sbox_layer(Fp* s, const uint32_t* __restrict__ rc) {
  uint32_t x[N], x2[N], x3[N];
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = (s[i] + Fp::raw(rc[i])).v;
  fp_mul_batch_raw<N>(x2, x, x);
  ZKSP_SCHED_FENCE();
  fp_mul_batch_raw<N>(x3, x2, x);
  ZKSP_SCHED_FENCE();
  fp_mul_batch_raw<N>(x2, x2, x2);
#pragma unroll
  for (int i = 0; i < N; ++i) x2[i] = fp_correct(x2[i]);
  ZKSP_SCHED_FENCE();
  fp_mul_batch_raw<N>(x, x3, x2);
#pragma unroll
  for (int i = 0; i < N; ++i) s[i] = Fp::raw(fp_correct(x[i]));
  ZKSP_SCHED_FENCE();
}

// circ(2*M4, M4, M4, M4) with M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]
ZKSP_HD void p2_external_linear(Fp* s) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    // 11 additions per 4x4 block (the evaluation order Plonky3's apply_mat4 uses)
    Fp x0 = s[4 * c], x1 = s[4 * c + 1], x2 = s[4 * c + 2], x3 = s[4 * c + 3];
    Fp t01 = x0 + x1, t23 = x2 + x3;
    Fp t0123 = t01 + t23;
    Fp t01123 = t0123 + x1, t01233 = t0123 + x3;
    s[4 * c + 3] = t01233 + x0.dbl();  // 3 x0 + x1 + x2 + 2 x3
    s[4 * c + 1] = t01123 + x2.dbl();  // x0 + 2 x1 + 3 x2 + x3
    s[4 * c] = t01123 + t01;           // 2 x0 + 3 x1 + x2 + x3
    s[4 * c + 2] = t01233 + t23;       // x0 + x1 + 2 x2 + 3 x3
  }
  Fp sums[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) sums[j] = (s[j] + s[4 + j]) + (s[8 + j] + s[12 + j]);
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = s[i] + sums[i & 3];
}

// y_i = d_i * x_i + sum(x), d = [-2, 1, 2, 4, ..., 8192, 32768]
ZKSP_HD void p2_internal_linear(Fp* s, const P2Consts* __restrict__ k) {
  Fp sum = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  sum = sum + (((s[8] + s[9]) + (s[10] + s[11])) + ((s[12] + s[13]) + (s[14] + s[15])));
  Fp d[13], prod[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) d[i] = Fp::raw(k->diag[3 + i]);
  fp_mul_batch<13>(prod, s + 3, d);
  s[0] = sum - s[0].dbl();   // d_0 = -2
  s[1] = sum + s[1];         // d_1 = 1
  s[2] = sum + s[2].dbl();   // d_2 = 2
#pragma unroll
  for (int i = 0; i < 13; ++i) s[3 + i] = prod[i] + sum;
}

ZKSP_HD void p2_permute(Fp* s, const P2Consts* __restrict__ k) {
  p2_external_linear(s);
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
    p2_sbox_layer<8>(s, k->ext[r]);
    p2_sbox_layer<8>(s + 8, k->ext[r] + 8);
    p2_external_linear(s);
  }
#pragma unroll 1
  for (int r = 0; r < 13; ++r) {
    s[0] = p2_sbox(s[0] + Fp::raw(k->internal[r]));
    p2_internal_linear(s, k);
  }
#pragma unroll 1
  for (int r = 4; r < 8; ++r) {
    p2_sbox_layer<8>(s, k->ext[r]);
    p2_sbox_layer<8>(s + 8, k->ext[r] + 8);
    p2_external_linear(s);
  }
}

}  // namespace zksp
