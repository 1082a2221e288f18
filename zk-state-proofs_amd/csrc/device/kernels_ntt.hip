// BabyBear radix-2 NTT / coset LDE (hot-path row a4; replaces p3-dft
// 0.1.4-succinct, reference Cargo.lock:5226, reached beneath
// prover/src/bin/main.rs:71-74).
//
// One workgroup owns one column and keeps it in LDS for the whole
// interpolate -> rescale -> 2x evaluate pipeline, so HBM sees exactly the
// algorithmic traffic: 4*H bytes read, 4*H (coefficients) + 8*H (LDE) written.
//   inverse:  decimation-in-frequency, natural -> bit-reversed (no permutation pass)
//   forward:  decimation-in-time,      bit-reversed -> natural
// Scale tables are stored pre-permuted by bit reversal, so coefficients never
// need to be reordered; consumers of `coefs_br` index it the same way.
//
// Butterflies run three radix-2 stages at a time on 8 values held in registers:
// a height-2^11 transform is 4 passes (3+3+3+2 stages) instead of 11, the first
// inverse pass reads HBM directly and the last forward pass writes HBM directly,
// both coalesced.  LDS rows are padded by 4 words per 32 so that the stride-4
// pass is bank-conflict free.
#include "kernels.h"

namespace zksp {

constexpr int kLdeThreads = 256;

__device__ __forceinline__ int lds_idx(int i) { return i + ((i >> 5) << 2); }

// One pass of R radix-2 stages, in registers.
//   DIF: stages s, s-1, .., s-R+1 (block sizes 2^s ..); group stride q = 2^(s-R)
//   DIT: stages s, s+1, .., s+R-1;                      group stride q = 2^(s-1)
// Sources / sinks are chosen per pass: LDS (padded) or global memory, with an
// optional element-wise scale before (pre) or after (post) the butterflies.
template <int R, bool DIF>
__device__ __forceinline__ void ntt_pass(const Fp* src_lds, const uint32_t* __restrict__ src_glb,
                                         const uint32_t* __restrict__ pre_scale, Fp* dst_lds,
                                         uint32_t* __restrict__ dst_glb, uint32_t* __restrict__ dst_glb2,
                                         const uint32_t* __restrict__ post_scale, const uint32_t* __restrict__ tw,
                                         int logh, int s, int tid) {
  constexpr int E = 1 << R;
  const int qlog = DIF ? s - R : s - 1;
  const int q = 1 << qlog;
  const int ngroups = (1 << logh) >> R;
  for (int g = tid; g < ngroups; g += kLdeThreads) {
    const int g_lo = g & (q - 1), g_hi = g >> qlog;
    const int base = (g_hi << (qlog + R)) | g_lo;
    Fp x[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int pos = base + (k << qlog);
      x[k] = src_glb ? Fp::raw(src_glb[pos]) : src_lds[lds_idx(pos)];
      if (pre_scale) x[k] = x[k] * Fp::raw(pre_scale[pos]);
    }
#pragma unroll
    for (int st = 0; st < R; ++st) {
      // DIF walks from the widest pairing (k, k + E/2) down, DIT from (k, k+1) up
      const int hk = DIF ? (E >> (st + 1)) : (1 << st);
      const int stage = DIF ? s - st : s + st;
      const int tw_shift = logh - stage;
#pragma unroll
      for (int k = 0; k < E; ++k) {
        if ((k & hk) != 0) continue;  // k is the lower element of its pair
        const int j = ((k & (hk - 1)) << qlog) + g_lo;
        const Fp w = Fp::raw(tw[j << tw_shift]);
        if (DIF) {
          Fp u = x[k], v = x[k + hk];
          x[k] = u + v;
          x[k + hk] = (u - v) * w;
        } else {
          Fp u = x[k], t = x[k + hk] * w;
          x[k] = u + t;
          x[k + hk] = u - t;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int pos = base + (k << qlog);
      Fp v = x[k];
      if (post_scale) v = v * Fp::raw(post_scale[pos]);
      if (dst_lds) dst_lds[lds_idx(pos)] = v;
      if (dst_glb) dst_glb[pos] = v.v;
      if (dst_glb2) dst_glb2[pos] = v.v;
    }
  }
}

template <bool DIF>
__device__ __forceinline__ void ntt_pass_r(int r, const Fp* src_lds, const uint32_t* src_glb, const uint32_t* pre_scale,
                                           Fp* dst_lds, uint32_t* dst_glb, uint32_t* dst_glb2,
                                           const uint32_t* post_scale, const uint32_t* tw, int logh, int s, int tid) {
  if (r == 3) ntt_pass<3, DIF>(src_lds, src_glb, pre_scale, dst_lds, dst_glb, dst_glb2, post_scale, tw, logh, s, tid);
  else if (r == 2) ntt_pass<2, DIF>(src_lds, src_glb, pre_scale, dst_lds, dst_glb, dst_glb2, post_scale, tw, logh, s, tid);
  else ntt_pass<1, DIF>(src_lds, src_glb, pre_scale, dst_lds, dst_glb, dst_glb2, post_scale, tw, logh, s, tid);
}

__global__ __launch_bounds__(kLdeThreads) void lde_lds_kernel(const uint32_t* __restrict__ in,
                                                              uint32_t* __restrict__ coefs_br,
                                                              uint32_t* __restrict__ out,
                                                              const uint32_t* __restrict__ tw_fwd,
                                                              const uint32_t* __restrict__ tw_inv,
                                                              const uint32_t* __restrict__ in_scale_br,
                                                              int scale_sel_shift, int scale_sel_mask,
                                                              const uint32_t* __restrict__ out_scale_br, int logh,
                                                              size_t ncols) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int h = 1 << logh;
  const int padded = h + (h >> 3) + 4;
  Fp* coef = reinterpret_cast<Fp*>(smem);
  Fp* work = coef + padded;
  const int tid = threadIdx.x;
  for (size_t col = blockIdx.x; col < ncols; col += gridDim.x) {
    const uint32_t* src = in + col * (size_t)h;
    // per-column choice of the input-coset scale table (quotient chunks differ)
    const uint32_t* isc = in_scale_br + (size_t)((col >> scale_sel_shift) & (size_t)scale_sel_mask) * h;
    uint32_t* cdst = coefs_br ? coefs_br + col * (size_t)h : nullptr;
    // ---- inverse transform: DIF, stages logh .. 1 ----
    {
      int s = logh;
      bool first = true;
      while (s > 0) {
        const int r = s >= 3 ? 3 : s;
        const bool last = (s - r) == 0;
        ntt_pass_r<true>(r, coef, first ? src : nullptr, nullptr, coef, nullptr, last ? cdst : nullptr,
                         last ? isc : nullptr, tw_inv, logh, s, tid);
        __syncthreads();
        s -= r;
        first = false;
      }
    }
    // ---- two forward transforms: DIT, stages 1 .. logh ----
    for (int cs = 0; cs < 2; ++cs) {
      const uint32_t* osc = out_scale_br + (size_t)cs * h;
      uint32_t* dst = out + (col * 2 + cs) * (size_t)h;
      int s = 1;
      bool first = true;
      while (s <= logh) {
        const int rem = logh - s + 1;
        const int r = first ? ((rem % 3) ? (rem % 3) : 3) : 3;
        const bool last = (s + r) > logh;
        ntt_pass_r<false>(r, first ? coef : work, nullptr, first ? osc : nullptr, last ? nullptr : work,
                          last ? dst : nullptr, nullptr, nullptr, tw_fwd, logh, s, tid);
        __syncthreads();
        s += r;
        first = false;
      }
    }
  }
}

void launch_lde(hipStream_t stream, const uint32_t* in, uint32_t* coefs_br, uint32_t* out, const uint32_t* tw_fwd,
                const uint32_t* tw_inv, const uint32_t* in_scale_br, int scale_sel_shift, int scale_sel_mask,
                const uint32_t* out_scale_br, int logh, size_t ncols) {
  if (ncols == 0) return;
  const size_t h = (size_t)1 << logh;
  size_t smem = 2 * sizeof(uint32_t) * (h + (h >> 3) + 4);
  size_t grid = ncols < 65536 ? ncols : 65536;
  hipLaunchKernelGGL(lde_lds_kernel, dim3((unsigned)grid), dim3(kLdeThreads), smem, stream, in, coefs_br, out, tw_fwd,
                     tw_inv, in_scale_br, scale_sel_shift, scale_sel_mask, out_scale_br, logh, ncols);
}

int lde_configure() {
  // the LDS-resident transform may use the whole 160 KiB of a CU
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(lde_lds_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace zksp
