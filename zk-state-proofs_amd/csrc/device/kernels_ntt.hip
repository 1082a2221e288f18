// BabyBear radix-2 NTT / coset LDE (hot-path row a4; replaces p3-dft
// 0.1.4-succinct, reference Cargo.lock:5226, reached beneath
// prover/src/bin/main.rs:71-74).
//
// One workgroup owns NC adjacent columns and keeps them in LDS for the whole
// interpolate -> rescale -> 2x evaluate pipeline, so HBM sees exactly the
// algorithmic traffic: 4*H bytes read, 4*H (coefficients) + 8*H (LDE) written
// per column.
//   inverse:  decimation-in-frequency, natural -> bit-reversed (no permutation pass)
//   forward:  decimation-in-time,      bit-reversed -> natural
// Scale tables are stored pre-permuted by bit reversal, so coefficients never
// need to be reordered; consumers of `coefs_br` index it the same way.
//
// Butterflies run three radix-2 stages at a time on 8 values held in registers:
// a height-2^11 transform is 4 passes (3+3+3+2 stages) instead of 11, the first
// inverse pass reads HBM directly and the last forward pass writes HBM directly,
// both coalesced.  LDS rows are padded by 4 words per 32 so that the stride-4
// pass is bank-conflict free.  A thread applies the same butterfly group to NC
// columns, which shares the index arithmetic and the twiddle loads (about half
// of the instruction stream) between them.
#include "kernels.h"

#include <cstdlib>

namespace zksp {

constexpr int kLdeThreads = 256;

__device__ __forceinline__ int lds_idx(int i) { return i + ((i >> 5) << 2); }

struct PassIo {
  const Fp* src_lds;             // padded LDS image of column 0 (nullptr: read global)
  const uint32_t* src_glb;       // column 0 in global memory
  const uint32_t* pre_scale;     // applied after the load (indexed by position)
  Fp* dst_lds;                   // padded LDS image of column 0 (nullptr: none)
  uint32_t* dst_glb;             // column 0 in global memory (nullptr: none)
  const uint32_t* post_scale;    // applied before the store
  size_t src_glb_stride, dst_glb_stride;  // words between adjacent columns
  int lds_stride;                // words between adjacent columns' LDS images
};

// One pass of R radix-2 stages, in registers, on NC columns at once.
//   DIF: stages s, s-1, .., s-R+1 (block sizes 2^s ..); group stride q = 2^(s-R)
//   DIT: stages s, s+1, .., s+R-1;                      group stride q = 2^(s-1)
template <int R, bool DIF, int NC>
__device__ __forceinline__ void ntt_pass(const PassIo& io, const uint32_t* __restrict__ tw, int logh, int s, int ncols,
                                         int tid) {
  constexpr int E = 1 << R;
  const int qlog = DIF ? s - R : s - 1;
  const int q = 1 << qlog;
  const int ngroups = (1 << logh) >> R;
  for (int g = tid; g < ngroups; g += kLdeThreads) {
    const int g_lo = g & (q - 1), g_hi = g >> qlog;
    const int base = (g_hi << (qlog + R)) | g_lo;
    Fp x[NC][E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int pos = base + (k << qlog);
      const int li = lds_idx(pos);
      Fp sc = Fp::one();
      if (io.pre_scale) sc = Fp::raw(io.pre_scale[pos]);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (c < ncols) {
          x[c][k] = io.src_glb ? Fp::raw(io.src_glb[(size_t)c * io.src_glb_stride + pos])
                               : io.src_lds[c * io.lds_stride + li];
          if (io.pre_scale) x[c][k] = x[c][k] * sc;
        } else {
          x[c][k] = Fp::zero();
        }
      }
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
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (DIF) {
            Fp u = x[c][k], v = x[c][k + hk];
            x[c][k] = u + v;
            x[c][k + hk] = (u - v) * w;
          } else {
            Fp u = x[c][k], t = x[c][k + hk] * w;
            x[c][k] = u + t;
            x[c][k + hk] = u - t;
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int pos = base + (k << qlog);
      const int li = lds_idx(pos);
      Fp sc = Fp::one();
      if (io.post_scale) sc = Fp::raw(io.post_scale[pos]);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (c >= ncols) continue;
        Fp v = x[c][k];
        if (io.post_scale) v = v * sc;
        if (io.dst_lds) io.dst_lds[c * io.lds_stride + li] = v;
        if (io.dst_glb) io.dst_glb[(size_t)c * io.dst_glb_stride + pos] = v.v;
      }
    }
  }
}

template <bool DIF, int NC>
__device__ __forceinline__ void ntt_pass_r(int r, const PassIo& io, const uint32_t* tw, int logh, int s, int ncols,
                                           int tid) {
  if (r == 3) ntt_pass<3, DIF, NC>(io, tw, logh, s, ncols, tid);
  else if (r == 2) ntt_pass<2, DIF, NC>(io, tw, logh, s, ncols, tid);
  else ntt_pass<1, DIF, NC>(io, tw, logh, s, ncols, tid);
}

template <int NC>
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
  Fp* coef = reinterpret_cast<Fp*>(smem);   // NC images
  Fp* work = coef + (size_t)NC * padded;    // NC images
  const int tid = threadIdx.x;
  const size_t ngroups = (ncols + NC - 1) / NC;
  for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const size_t col = grp * NC;
    const int nc = (int)((ncols - col) < (size_t)NC ? (ncols - col) : (size_t)NC);
    const uint32_t* src = in + col * (size_t)h;
    // the NC columns of a group share the input-coset scale table (NC divides the
    // 4-column quotient chunks; trace columns all use table 0)
    const uint32_t* isc = in_scale_br + (size_t)((col >> scale_sel_shift) & (size_t)scale_sel_mask) * h;
    uint32_t* cdst = coefs_br ? coefs_br + col * (size_t)h : nullptr;
    // ---- inverse transform: DIF, stages logh .. 1 ----
    {
      int s = logh;
      bool first = true;
      while (s > 0) {
        const int r = s >= 3 ? 3 : s;
        const bool last = (s - r) == 0;
        PassIo io;
        io.src_lds = coef;
        io.src_glb = first ? src : nullptr;
        io.pre_scale = nullptr;
        io.dst_lds = coef;
        io.dst_glb = last ? cdst : nullptr;
        io.post_scale = last ? isc : nullptr;
        io.src_glb_stride = (size_t)h;
        io.dst_glb_stride = (size_t)h;
        io.lds_stride = padded;
        ntt_pass_r<true, NC>(r, io, tw_inv, logh, s, nc, tid);
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
        PassIo io;
        io.src_lds = first ? coef : work;
        io.src_glb = nullptr;
        io.pre_scale = first ? osc : nullptr;
        io.dst_lds = last ? nullptr : work;
        io.dst_glb = last ? dst : nullptr;
        io.post_scale = nullptr;
        io.src_glb_stride = 0;
        io.dst_glb_stride = (size_t)2 * h;  // adjacent columns are 2 cosets apart in the LDE
        io.lds_stride = padded;
        ntt_pass_r<false, NC>(r, io, tw_fwd, logh, s, nc, tid);
        __syncthreads();
        s += r;
        first = false;
      }
    }
  }
}

// columns per workgroup: 2 while two padded double images fit comfortably in LDS
static int lde_cols_per_block(int logh) {
  if (const char* e = getenv("ZKSP_LDE_NC")) return atoi(e) == 4 && logh <= 11 ? 4 : (atoi(e) == 2 && logh <= 12 ? 2 : 1);
  return logh <= 12 ? 2 : 1;
}

void launch_lde(hipStream_t stream, const uint32_t* in, uint32_t* coefs_br, uint32_t* out, const uint32_t* tw_fwd,
                const uint32_t* tw_inv, const uint32_t* in_scale_br, int scale_sel_shift, int scale_sel_mask,
                const uint32_t* out_scale_br, int logh, size_t ncols) {
  if (ncols == 0) return;
  const size_t h = (size_t)1 << logh;
  const int nc = lde_cols_per_block(logh);
  const size_t smem = (size_t)nc * 2 * sizeof(uint32_t) * (h + (h >> 3) + 4);
  const size_t groups = (ncols + nc - 1) / nc;
  const unsigned grid = (unsigned)(groups < 65536 ? groups : 65536);
  if (nc == 4)
    hipLaunchKernelGGL(lde_lds_kernel<4>, dim3(grid), dim3(kLdeThreads), smem, stream, in, coefs_br, out, tw_fwd, tw_inv,
                       in_scale_br, scale_sel_shift, scale_sel_mask, out_scale_br, logh, ncols);
  else if (nc == 2)
    hipLaunchKernelGGL(lde_lds_kernel<2>, dim3(grid), dim3(kLdeThreads), smem, stream, in, coefs_br, out, tw_fwd, tw_inv,
                       in_scale_br, scale_sel_shift, scale_sel_mask, out_scale_br, logh, ncols);
  else
    hipLaunchKernelGGL(lde_lds_kernel<1>, dim3(grid), dim3(kLdeThreads), smem, stream, in, coefs_br, out, tw_fwd, tw_inv,
                       in_scale_br, scale_sel_shift, scale_sel_mask, out_scale_br, logh, ncols);
}

int lde_configure() {
  // the LDS-resident transform may use the whole 160 KiB of a CU
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lde_lds_kernel<1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(lde_lds_kernel<4>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(lde_lds_kernel<2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace zksp
