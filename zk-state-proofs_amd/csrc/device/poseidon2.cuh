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
// instruction order pinned: hipcc otherwise emits each product as one serial
// mad -> mul_lo -> mad -> add -> min chain through a shared temporary, which
// leaves a wave stalled on its own previous result at the 4-5 waves per SIMD
// these kernels run at.  N independent chains per stage cover that latency.
template <int N>
ZKSP_HD void fp_mul_batch(Fp* out, const Fp* a, const Fp* b) {
  uint64_t t[N];
  uint32_t m[N];
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = (uint64_t)a[i].v * b[i].v;
  ZKSP_SCHED_FENCE();
#pragma unroll
  for (int i = 0; i < N; ++i) m[i] = (uint32_t)t[i] * kMontyNegMu;
  ZKSP_SCHED_FENCE();
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = t[i] + (uint64_t)m[i] * kP;
  ZKSP_SCHED_FENCE();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    uint32_t r = (uint32_t)(t[i] >> 32), r2 = r - kP;
    out[i] = Fp::raw(r < r2 ? r : r2);
  }
  ZKSP_SCHED_FENCE();
}

ZKSP_HD Fp p2_sbox(Fp x) {
  Fp x2 = x.sqr(), x3 = x2 * x, x4 = x2.sqr();
  return x3 * x4;
}

// x -> (x + rc)^7 on N lanes
template <int N>
ZKSP_HD void p2_sbox_layer(Fp* s, const uint32_t* __restrict__ rc) {
  Fp x2[N], x3[N];
#pragma unroll
  for (int i = 0; i < N; ++i) s[i] = s[i] + Fp::raw(rc[i]);
  fp_mul_batch<N>(x2, s, s);
  fp_mul_batch<N>(x3, x2, s);
  fp_mul_batch<N>(x2, x2, x2);
  fp_mul_batch<N>(s, x3, x2);
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
  Fp d[16], prod[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) d[i] = Fp::raw(k->diag[i]);
  fp_mul_batch<8>(prod, s, d);
  fp_mul_batch<8>(prod + 8, s + 8, d + 8);
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = prod[i] + sum;
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
