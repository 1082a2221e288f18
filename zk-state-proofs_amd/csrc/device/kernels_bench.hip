// Instruction-rate probes behind zksp_hip_microbench(): they price the integer
// ALU roofline that bounds the Poseidon2 kernels (DESIGN.md "ALU roofline").
#include "kernels.h"

namespace zksp {

template <int WHICH>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x, c = a ^ b, d = a + 7;
  double fa = a, fb = b, fc = c, fd = d;
  uint64_t qa = a, qb = b, qc = c, qd = d;
  Fp ma = Fp::raw(a % kP), mb = Fp::raw(b % kP), mc = Fp::raw(c % kP), md = Fp::raw(d % kP);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (WHICH == 0) { a += b; b += c; c += d; d += a; }
      if (WHICH == 1) { a *= b; b *= c; c *= d; d *= a; }
      if (WHICH == 2) { a = __umulhi(a, b); b = __umulhi(b, c) | 1; c = __umulhi(c, d) | 3; d = __umulhi(d, a) | 5; }
      if (WHICH == 3) {
        qa = (uint64_t)(uint32_t)qa * (uint32_t)qb + qc;
        qb = (uint64_t)(uint32_t)qb * (uint32_t)qc + qd;
        qc = (uint64_t)(uint32_t)qc * (uint32_t)qd + qa;
        qd = (uint64_t)(uint32_t)qd * (uint32_t)qa + qb;
      }
      if (WHICH == 4) { ma = ma * mb; mb = mb * mc; mc = mc * md; md = md * ma; }
      if (WHICH == 5) { fa = fa * fb + fc; fb = fb * fc + fd; fc = fc * fd + fa; fd = fd * fa + fb; }
      if (WHICH == 6) { qa += qb << 1; qb += qc << 1; qc += qd << 1; qd += qa << 1; }  // v_lshl_add_u64
    }
  }
  uint32_t r = a ^ b ^ c ^ d ^ (uint32_t)(fa + fb + fc + fd) ^ (uint32_t)(qa ^ qb ^ qc ^ qd) ^ ma.v ^ mb.v ^ mc.v ^ md.v;
  if (r == 0x12345678u) out[0] = r;  // keeps the chains alive
}

// Poseidon2 permutation throughput at a chosen residency: every thread runs
// `iters` dependent permutations on a register-resident state.
__global__ __launch_bounds__(256) void perm_rate_kernel(uint32_t* out, int iters, const P2Consts* __restrict__ k) {
  Fp s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = Fp::raw((threadIdx.x * 16 + i + blockIdx.x) % kP);
  for (int it = 0; it < iters; ++it) p2_permute(s, k);
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r ^= s[i].v;
  if (r == 0x12345678u) out[0] = r;
}
void launch_perm_rate_kernel(hipStream_t stream, uint32_t* out, int blocks, int iters, const P2Consts* k) {
  hipLaunchKernelGGL(perm_rate_kernel, dim3(blocks), dim3(256), 0, stream, out, iters, k);
}

void launch_rate_kernel(hipStream_t stream, int which, uint32_t* out, int blocks, int iters) {
  dim3 g(blocks), b(256);
  switch (which) {
    case 0: hipLaunchKernelGGL(rate_kernel<0>, g, b, 0, stream, out, 12345u, iters); break;
    case 1: hipLaunchKernelGGL(rate_kernel<1>, g, b, 0, stream, out, 12345u, iters); break;
    case 2: hipLaunchKernelGGL(rate_kernel<2>, g, b, 0, stream, out, 12345u, iters); break;
    case 3: hipLaunchKernelGGL(rate_kernel<3>, g, b, 0, stream, out, 12345u, iters); break;
    case 4: hipLaunchKernelGGL(rate_kernel<4>, g, b, 0, stream, out, 12345u, iters); break;
    case 5: hipLaunchKernelGGL(rate_kernel<5>, g, b, 0, stream, out, 12345u, iters); break;
    default: hipLaunchKernelGGL(rate_kernel<6>, g, b, 0, stream, out, 12345u, iters); break;
  }
}

}  // namespace zksp
