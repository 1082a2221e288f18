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
#include "kernels.h"

namespace zksp {

constexpr int kLdeThreads = 256;

// In-LDS radix-2 butterflies, one barrier per stage.
template <bool INVERSE>
__device__ __forceinline__ void lds_ntt(Fp* buf, const uint32_t* __restrict__ tw, int logh, int tid) {
  const int half_n = 1 << (logh - 1);
  if (INVERSE) {
    // DIF: stage sizes m = 2^s, s = logh .. 1
    for (int s = logh; s >= 1; --s) {
      const int half = 1 << (s - 1);
      const int tw_step = logh - s;  // w_m^j = w_H^(j << (logh - s))
      for (int b = tid; b < half_n; b += kLdeThreads) {
        int j = b & (half - 1);
        int i0 = ((b >> (s - 1)) << s) | j;
        int i1 = i0 + half;
        Fp u = buf[i0], v = buf[i1];
        buf[i0] = u + v;
        buf[i1] = (u - v) * Fp::raw(tw[j << tw_step]);
      }
      __syncthreads();
    }
  } else {
    for (int s = 1; s <= logh; ++s) {
      const int half = 1 << (s - 1);
      const int tw_step = logh - s;
      for (int b = tid; b < half_n; b += kLdeThreads) {
        int j = b & (half - 1);
        int i0 = ((b >> (s - 1)) << s) | j;
        int i1 = i0 + half;
        Fp u = buf[i0], t = buf[i1] * Fp::raw(tw[j << tw_step]);
        buf[i0] = u + t;
        buf[i1] = u - t;
      }
      __syncthreads();
    }
  }
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
  Fp* coef = reinterpret_cast<Fp*>(smem);
  Fp* work = coef + h;
  const int tid = threadIdx.x;
  for (size_t col = blockIdx.x; col < ncols; col += gridDim.x) {
    const uint32_t* src = in + col * (size_t)h;
    for (int i = tid; i < h; i += kLdeThreads) coef[i] = Fp::raw(src[i]);
    __syncthreads();
    lds_ntt<true>(coef, tw_inv, logh, tid);
    // per-column choice of the input-coset scale table (quotient chunks differ)
    const uint32_t* isc = in_scale_br + (size_t)((col >> scale_sel_shift) & (size_t)scale_sel_mask) * h;
    for (int i = tid; i < h; i += kLdeThreads) {
      Fp c = coef[i] * Fp::raw(isc[i]);
      coef[i] = c;
      if (coefs_br) coefs_br[col * (size_t)h + i] = c.v;
    }
    __syncthreads();
    for (int cs = 0; cs < 2; ++cs) {
      const uint32_t* sc = out_scale_br + (size_t)cs * h;
      for (int i = tid; i < h; i += kLdeThreads) work[i] = coef[i] * Fp::raw(sc[i]);
      __syncthreads();
      lds_ntt<false>(work, tw_fwd, logh, tid);
      uint32_t* dst = out + (col * 2 + cs) * (size_t)h;
      for (int i = tid; i < h; i += kLdeThreads) dst[i] = work[i].v;
      __syncthreads();
    }
  }
}

void launch_lde(hipStream_t stream, const uint32_t* in, uint32_t* coefs_br, uint32_t* out, const uint32_t* tw_fwd,
                const uint32_t* tw_inv, const uint32_t* in_scale_br, int scale_sel_shift, int scale_sel_mask,
                const uint32_t* out_scale_br, int logh, size_t ncols) {
  if (ncols == 0) return;
  size_t smem = (size_t)2 * sizeof(uint32_t) << logh;
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
