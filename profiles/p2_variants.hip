// Measurement tool (not part of the product): Poseidon2 permutation rate on a register-resident state for the
// variants of the linear layers (MODE 0: the product's; bit 0: external layers in two 32-bit planes, bit 1: internal
// rounds in two 32-bit planes, 4: round 2's external layer), and the issue rate of every opcode
// class the permutation uses.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include profiles/p2_variants.hip -o gpurun_out/p2_variants && gpurun_out/p2_variants
// Constants are pseudo-random words below p (timing does not depend on their values); all variants must print the
// same checksum.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "../zk-state-proofs_amd/csrc/device/poseidon2.hpp"

using namespace zksp;


// ---- variants of the linear layers (the product's are p2s_external_linear / p2s_internal_round in poseidon2.hpp) ----
// (1) two 32-bit planes instead of 64-bit accumulators: W = the layer over the words (wrapping, = y mod 2^32),
//     H = the layer over x >> 16 (an estimate of y / 2^16 good to 36), q = (273 H + 2^22) >> 23, t = W - q p.
__device__ __forceinline__ int32_t reduce_planes(uint32_t w, int32_t h) {
  const int32_t q = (__mul24(h, 273) + (1 << 22)) >> 23;
  return (int32_t)(w - (uint32_t)q * kP);
}
__device__ __forceinline__ void mix_plane(uint32_t* x, const uint32_t* __restrict__ rc, int shift) {
  uint32_t z[16];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t x0 = x[4 * c], x1 = x[4 * c + 1], x2 = x[4 * c + 2], x3 = x[4 * c + 3];
    const uint32_t t01 = x0 + x1, t23 = x2 + x3, sum = t01 + t23;
    z[4 * c] = sum + t01 + x1;
    z[4 * c + 1] = sum + x1 + (x2 << 1);
    z[4 * c + 2] = sum + t23 + x3;
    z[4 * c + 3] = sum + x3 + (x0 << 1);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t col = (z[j] + z[4 + j]) + (z[8 + j] + z[12 + j]);
#pragma unroll
    for (int c = 0; c < 4; ++c) x[4 * c + j] = z[4 * c + j] + col + (rc[4 * c + j] >> shift);
  }
}
__device__ __forceinline__ void ext_planes(int32_t* s, const uint32_t* __restrict__ rc) {
  uint32_t w[16], h[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { w[i] = (uint32_t)s[i]; h[i] = (uint32_t)(s[i] >> 16); }
  mix_plane(w, rc, 0);
  mix_plane(h, rc, 16);
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = reduce_planes(w[i], (int32_t)h[i]);
}
template <bool FULL>
__device__ __forceinline__ void int_planes(int32_t* s, const P2Consts* __restrict__ k, const int64_t* __restrict__ add) {
  s[0] = p2s_sbox(s[0]);
  uint32_t w = 0, h = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { w += (uint32_t)s[i]; h += (uint32_t)(s[i] >> 16); }
  const int64_t sr = (int64_t)reduce_planes(w, (int32_t)h) * (int64_t)kRModP;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int64_t t = (int64_t)s[i] * (int64_t)k->sdiag[i] + sr;
    if (FULL || i == 0) t += add[FULL ? i : 0];
    s[i] = fps_redc(t);
  }
}
// (2) the external layer as it was in round 2: eleven 64-bit additions per 4-block, the round constant inside the 64-bit sum
__device__ __forceinline__ void ext_wide_r2(int32_t* s, const uint32_t* __restrict__ rc) {
  int64_t y[16];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t x0 = s[4 * c], x1 = s[4 * c + 1], x2 = s[4 * c + 2], x3 = s[4 * c + 3];
    const int64_t sum = (x0 + x1) + (x2 + x3);
    y[4 * c] = sum + x0 + 2 * x1;
    y[4 * c + 1] = sum + x1 + 2 * x2;
    y[4 * c + 2] = sum + x2 + 2 * x3;
    y[4 * c + 3] = sum + x3 + 2 * x0;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t col = (y[j] + y[4 + j]) + (y[8 + j] + y[12 + j]);
#pragma unroll
    for (int c = 0; c < 4; ++c) s[4 * c + j] = fps_reduce_small(y[4 * c + j] + col + (int64_t)rc[4 * c + j]);
  }
}

template <int MODE>
__device__ __forceinline__ void permute_variant(int32_t* s, const P2Consts* __restrict__ k) {
  auto ext = [&](const uint32_t* rc) {
    if (MODE & 4) ext_wide_r2(s, rc);
    else if (MODE & 1) ext_planes(s, rc);
    else p2s_external_linear(s, rc);
  };
  ext(k->lin_rc[0]);
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = p2s_sbox(s[i]);
    ext(k->lin_rc[r + 1]);
  }
#pragma unroll 1
  for (int r = 0; r < 12; ++r) {
    if (MODE & 2) int_planes<false>(s, k, &k->int_add[r]); else p2s_internal_round<false>(s, k, &k->int_add[r]);
  }
  if (MODE & 2) int_planes<true>(s, k, k->int_last); else p2s_internal_round<true>(s, k, k->int_last);
#pragma unroll 1
  for (int r = 4; r < 8; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = p2s_sbox(s[i]);
    ext(k->lin_rc[r + 1]);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void perm_kernel(uint32_t* out, int iters, const P2Consts* __restrict__ k) {
  int32_t s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = (int32_t)((threadIdx.x * 16 + i + blockIdx.x * 4096u) % kP);
  for (int it = 0; it < iters; ++it) permute_variant<MODE>(s, k);
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r ^= fps_canon(s[i]) * (2 * i + 1);
  atomicXor(out, r);
}


// ---- per-opcode issue rates (inline asm so the compiler cannot merge or strength-reduce the chains) ----
#define PROBE_KERNEL(NAME, ASM, CONSTRAINT_A, CONSTRAINT_B, TYPE)                                                   \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters) {                                           \
    TYPE a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    TYPE b = blockIdx.x + 3;                                                                                        \
    uint32_t b32 = blockIdx.x + 5;                                                                                  \
    for (int i = 0; i < iters; ++i) {                                                                               \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                               \
        asm volatile(ASM : CONSTRAINT_A(a0) : CONSTRAINT_B(b), "v"(b32) : "vcc");                                                     \
        asm volatile(ASM : CONSTRAINT_A(a1) : CONSTRAINT_B(b), "v"(b32) : "vcc");                                                     \
        asm volatile(ASM : CONSTRAINT_A(a2) : CONSTRAINT_B(b), "v"(b32) : "vcc");                                                     \
        asm volatile(ASM : CONSTRAINT_A(a3) : CONSTRAINT_B(b), "v"(b32) : "vcc");                                                     \
        asm volatile(ASM : CONSTRAINT_A(a4) : CONSTRAINT_B(b), "v"(b32) : "vcc");                                                     \
        asm volatile(ASM : CONSTRAINT_A(a5) : CONSTRAINT_B(b), "v"(b32) : "vcc");                                                     \
        asm volatile(ASM : CONSTRAINT_A(a6) : CONSTRAINT_B(b), "v"(b32) : "vcc");                                                     \
        asm volatile(ASM : CONSTRAINT_A(a7) : CONSTRAINT_B(b), "v"(b32) : "vcc");                                                     \
      }                                                                                                             \
    }                                                                                                               \
    TYPE r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                 \
    if ((uint32_t)r == 0x12345678u) out[1] = (uint32_t)r;                                                           \
  }
#define RW "+v"
#define RO "v"
PROBE_KERNEL(k_add_u32, "v_add_u32_e32 %0, %0, %1", RW, RO, uint32_t)
PROBE_KERNEL(k_add3_u32, "v_add3_u32 %0, %0, %1, %1", RW, RO, uint32_t)
PROBE_KERNEL(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %1", RW, RO, uint32_t)
PROBE_KERNEL(k_add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1", RW, RO, uint32_t)
PROBE_KERNEL(k_ashr, "v_ashrrev_i32_e32 %0, 1, %0", RW, RO, uint32_t)
PROBE_KERNEL(k_xor, "v_xor_b32_e32 %0, %0, %1", RW, RO, uint32_t)
PROBE_KERNEL(k_min_u32, "v_min_u32_e32 %0, %0, %1", RW, RO, uint32_t)
PROBE_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1", RW, RO, uint32_t)
PROBE_KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1", RW, RO, uint32_t)
PROBE_KERNEL(k_mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %1", RW, RO, uint32_t)
PROBE_KERNEL(k_mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %1", RW, RO, uint32_t)
PROBE_KERNEL(k_mad_i32_i16_hi, "v_mad_i32_i16 %0, %0, %1, %0 op_sel:[1,0,0,0]", RW, RO, uint32_t)
PROBE_KERNEL(k_dot2_i32_i16, "v_dot2_i32_i16 %0, %0, %1, %0", RW, RO, uint32_t)
PROBE_KERNEL(k_perm_b32, "v_perm_b32 %0, %0, %1, %1", RW, RO, uint32_t)
PROBE_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7", RW, RO, uint32_t)
PROBE_KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1", RW, RO, uint32_t)
PROBE_KERNEL(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1", RW, RO, uint32_t)
PROBE_KERNEL(k_mad_i64_i32, "v_mad_i64_i32 %0, vcc, %2, %2, %0", RW, RO, uint64_t)
PROBE_KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %2, %2, %0", RW, RO, uint64_t)
PROBE_KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %0", RW, RO, uint64_t)
PROBE_KERNEL(k_fma_f64, "v_fma_f64 %0, %0, %0, %0", RW, RO, double)
PROBE_KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %0", RW, RO, float)
PROBE_KERNEL(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %0, %0", RW, RO, double)

template <typename K>
static void probe(const char* name, K kern, uint32_t* d_out) {
  const int blocks = 256 * 8, iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 10);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, iters);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("%-18s %7.2f T lane-ops/s\n", name, (double)blocks * 256.0 * iters * 64.0 / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

template <int MODE>
static void run(uint32_t* d_out, const P2Consts* d_k, int per_cu) {
  const int blocks = 256 * per_cu, iters = 64;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipMemset(d_out, 0, 4);
  hipLaunchKernelGGL(perm_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 2, d_k);
  uint32_t chk = 0;
  hipMemcpy(&chk, d_out, 4, hipMemcpyDeviceToHost);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(perm_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, d_k);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  printf("mode %d  %d workgroups/CU  %.3f Gperm/s  checksum %08x\n", MODE, per_cu, (double)blocks * 256.0 * iters / (best * 1e-3) / 1e9, chk);
  fflush(stdout);
}

int main() {
  P2Consts k;
  uint64_t x = 88172645463325252ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x % kP); };
  for (int r = 0; r < 8; ++r) for (int i = 0; i < 16; ++i) k.ext[r][i] = rnd();
  for (int r = 0; r < 13; ++r) k.internal[r] = rnd();
  for (int i = 0; i < 16; ++i) { k.diag[i] = rnd(); k.sdiag[i] = fps_centre(k.diag[i]); }
  for (int l = 0; l < 9; ++l) for (int i = 0; i < 16; ++i) { k.lin_rc[l][i] = rnd(); k.lin_add[l][i] = rnd(); }
  for (int r = 0; r < 13; ++r) k.int_add[r] = rnd();
  for (int i = 0; i < 16; ++i) k.int_last[i] = rnd();
  P2Consts* d_k;
  uint32_t* d_out;
  hipMalloc(&d_k, sizeof(k));
  hipMalloc(&d_out, 64);
  hipMemcpy(d_k, &k, sizeof(k), hipMemcpyHostToDevice);
#define P(k) probe(#k, k, d_out)
  P(k_add_u32); P(k_add3_u32); P(k_lshl_add_u32); P(k_add_sdwa); P(k_ashr); P(k_xor); P(k_min_u32); P(k_mul_lo); P(k_mul_hi);
  P(k_mad_i32_i24); P(k_mad_u32_u16); P(k_mad_i32_i16_hi); P(k_dot2_i32_i16); P(k_perm_b32); P(k_alignbit); P(k_pk_add_u16);
  P(k_pk_mul_lo_u16); P(k_mad_i64_i32); P(k_mad_u64_u32); P(k_lshl_add_u64); P(k_fma_f64); P(k_fma_f32); P(k_pk_fma_f32);
  for (int per_cu : {4, 8}) {
    run<0>(d_out, d_k, per_cu);
    run<1>(d_out, d_k, per_cu);
    run<2>(d_out, d_k, per_cu);
    run<3>(d_out, d_k, per_cu);
    run<4>(d_out, d_k, per_cu);
  }
  return 0;
}
