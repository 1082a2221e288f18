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

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace zksp {

constexpr int kLdeThreads = 256;

__device__ __forceinline__ int lds_idx(int i) { return i + ((i >> 5) << 2); }

struct PassIo {
  const Fp* src_lds;             // padded LDS image of column 0 (used unless kSrcGlb)
  const uint32_t* src_glb;       // column 0 in global memory (kSrcGlb)
  const uint32_t* pre_scale;     // applied after the load, indexed by position (kPre)
  Fp* dst_lds;                   // padded LDS image of column 0 (kDstLds)
  uint32_t* dst_glb;             // column 0 in global memory (kDstGlb)
  const uint32_t* post_scale;    // applied before the store (kPost)
  size_t src_glb_stride, dst_glb_stride;  // words between adjacent columns
  int lds_stride;                // words between adjacent columns' LDS images
};

// Which of PassIo's endpoints a pass uses.  Compile-time, so that the unrolled element
// loops carry no branches and LDS accesses are ds_* instructions rather than flat ones.
constexpr int kSrcGlb = 1, kPre = 2, kDstLds = 4, kDstGlb = 8, kPost = 16;

// One pass of R radix-2 stages, in registers, on NC columns at once.
//   DIF: stages s, s-1, .., s-R+1 (block sizes 2^s ..); group stride q = 2^(s-R)
//   DIT: stages s, s+1, .., s+R-1;                      group stride q = 2^(s-1)
template <int R, bool DIF, int NC, int F, bool FULL>
__device__ __forceinline__ void ntt_pass(const PassIo& io, const uint32_t* __restrict__ tw, int logh, int s,
                                         int ncols, int tid, int g_step = kLdeThreads) {
  constexpr int E = 1 << R;
  const int qlog = DIF ? s - R : s - 1;
  const int q = 1 << qlog;
  const int ngroups = (1 << logh) >> R;
  for (int g = tid; g < ngroups; g += g_step) {
    const int g_lo = g & (q - 1), g_hi = g >> qlog;
    const int base = (g_hi << (qlog + R)) | g_lo;
    Fp x[NC][E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int pos = base + (k << qlog);
      const int li = lds_idx(pos);
      Fp sc = Fp::one();
      if (F & kPre) sc = Fp::raw(io.pre_scale[pos]);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (FULL || c < ncols) {
          if (F & kSrcGlb) x[c][k] = Fp::raw(io.src_glb[(size_t)c * io.src_glb_stride + pos]);
          else x[c][k] = io.src_lds[c * io.lds_stride + li];
          if (F & kPre) x[c][k] = x[c][k] * sc;
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
#pragma unroll
      for (int k = 0; k < E; ++k) {
        if ((k & hk) != 0) continue;  // k is the lower element of its pair
        const int j = ((k & (hk - 1)) << qlog) + g_lo;
        const Fp w = Fp::raw(tw[(1 << (stage - 1)) + j]);  // per-stage table: consecutive across lanes
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (DIF) {
            // the difference goes into the product as a signed word: subtract (1), centred Montgomery product (3), canonical
            // (2) - nine instructions a butterfly with the sum's three, where (u - v) * w on residues takes eleven.  |u - v| < p
            // and w < p keep the product inside fps_redc's range and its result inside (-p, p).  (DESIGN.md section 4 "NTT, round 5")
            Fp u = x[c][k], v = x[c][k + hk];
            x[c][k] = u + v;
            x[c][k + hk] = Fp::raw(fps_canon(fps_mul((int32_t)u.v - (int32_t)v.v, (int32_t)w.v)));
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
      if (F & kPost) sc = Fp::raw(io.post_scale[pos]);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (!FULL && c >= ncols) continue;
        Fp v = x[c][k];
        if (F & kPost) v = v * sc;
        if (F & kDstLds) io.dst_lds[c * io.lds_stride + li] = v;
        if (F & kDstGlb) io.dst_glb[(size_t)c * io.dst_glb_stride + pos] = v.v;
      }
    }
  }
}

template <bool DIF, int NC, int F>
__device__ __forceinline__ void ntt_pass_f(int r, const PassIo& io, const uint32_t* tw, int logh, int s, int ncols,
                                           int tid) {
  if (ncols == NC) {
    if (r == 3) ntt_pass<3, DIF, NC, F, true>(io, tw, logh, s, ncols, tid);
    else if (r == 2) ntt_pass<2, DIF, NC, F, true>(io, tw, logh, s, ncols, tid);
    else ntt_pass<1, DIF, NC, F, true>(io, tw, logh, s, ncols, tid);
  } else {
    if (r == 3) ntt_pass<3, DIF, NC, F, false>(io, tw, logh, s, ncols, tid);
    else if (r == 2) ntt_pass<2, DIF, NC, F, false>(io, tw, logh, s, ncols, tid);
    else ntt_pass<1, DIF, NC, F, false>(io, tw, logh, s, ncols, tid);
  }
}

// Runtime flags -> the compile-time variant, over the list of combinations a caller can
// produce (the flags are uniform, so this is a scalar branch chain).  A combination outside
// the list is a programming error and traps.
template <bool DIF, int NC, int F0, int... Fs>
__device__ __forceinline__ void ntt_pass_sel(int flags, int r, const PassIo& io, const uint32_t* tw, int logh,
                                             int s, int ncols, int tid) {
  if (flags == F0) ntt_pass_f<DIF, NC, F0>(r, io, tw, logh, s, ncols, tid);
  else if constexpr (sizeof...(Fs) > 0) ntt_pass_sel<DIF, NC, Fs...>(flags, r, io, tw, logh, s, ncols, tid);
  else __builtin_trap();
}

__device__ __forceinline__ int pass_flags(const PassIo& io) {
  return (io.src_glb ? kSrcGlb : 0) | (io.pre_scale ? kPre : 0) | (io.dst_lds ? kDstLds : 0) |
         (io.dst_glb ? kDstGlb : 0) | (io.post_scale ? kPost : 0);
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
        io.dst_glb = last ? cdst : nullptr;  // cdst may be null: the caller does not want coefficients
        io.post_scale = last ? isc : nullptr;
        io.src_glb_stride = (size_t)h;
        io.dst_glb_stride = (size_t)h;
        io.lds_stride = padded;
        ntt_pass_sel<true, NC, kDstLds, kSrcGlb | kDstLds, kDstLds | kDstGlb | kPost, kDstLds | kPost,
                     kSrcGlb | kDstLds | kDstGlb | kPost, kSrcGlb | kDstLds | kPost>(pass_flags(io), r, io, tw_inv, logh,
                                                                                    s, nc, tid);
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
        ntt_pass_sel<false, NC, kDstLds, kPre | kDstLds, kDstGlb, kPre | kDstGlb>(pass_flags(io), r, io, tw_fwd, logh,
                                                                                  s, nc, tid);
        __syncthreads();
        s += r;
        first = false;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Heights 2^9 .. 2^11 (the bench workload is 2^11): ONE LDS image per column.
//
// The last inverse pass and the first forward pass both have unit group stride and the same
// radix RB = ((logh - 1) mod 3) + 1, so they work on the same groups of 2^RB consecutive
// elements, and at these heights a thread owns at most 16 of them per NC = 2 columns.  The
// coefficients therefore stay in registers across the boundary, for both cosets: the workspace
// image of lde_lds_kernel disappears, workgroups per CU double (LDS was the limiter: 4 -> 8), the
// last inverse pass writes no LDS and the two first forward passes read none.  The height is a
// template parameter, so every stage number, stride and padded LDS offset below is a constant.
// ---------------------------------------------------------------------------
// Where the NC columns of a group and the two cosets of the result lie in global memory (words).  A whole
// column of height H uses {H, H, 2H, H}; a 2^LOGH chunk of a taller column (lde_chunk_fixed_kernel) has
// its cosets a full column apart.
struct LdeGeom {
  size_t src_col, coef_col, dst_col, coset;
};

template <int LOGH, int NC, bool FULL>
__device__ __forceinline__ void lde_fixed_group(const uint32_t* __restrict__ src, uint32_t* __restrict__ cdst,
                                                uint32_t* __restrict__ dst, const uint32_t* __restrict__ tw_fwd,
                                                const uint32_t* __restrict__ tw_inv, const uint32_t* __restrict__ isc,
                                                const uint32_t* __restrict__ out_scale_br, Fp* buf, int nc, int tid,
                                                const LdeGeom geom) {
  constexpr int H = 1 << LOGH, PADDED = H + (H >> 3) + 4;
  constexpr int RB = ((LOGH - 1) % 3) + 1, EB = 1 << RB, NGB = H >> RB;
  constexpr int ITERS = NGB > kLdeThreads ? NGB / kLdeThreads : 1;
  static_assert((LOGH - RB) % 3 == 0 && LOGH - RB >= 3, "pass schedule");
  static_assert(ITERS * NC * EB <= 32, "coefficients held per thread");
  PassIo io;
  io.src_lds = buf;
  io.dst_lds = buf;
  io.src_glb = src;
  io.dst_glb = nullptr;
  io.pre_scale = io.post_scale = nullptr;
  io.src_glb_stride = geom.src_col;
  io.dst_glb_stride = geom.dst_col;  // adjacent columns are 2 cosets apart in the LDE
  io.lds_stride = PADDED;
  // ---- inverse transform, all passes but the last: DIF stages LOGH .. RB + 1 ----
  ntt_pass<3, true, NC, kSrcGlb | kDstLds, FULL>(io, tw_inv, LOGH, LOGH, nc, tid);
  __syncthreads();
#pragma unroll
  for (int s = LOGH - 3; s > RB; s -= 3) {
    ntt_pass<3, true, NC, kDstLds, FULL>(io, tw_inv, LOGH, s, nc, tid);
    __syncthreads();
  }
  // ---- boundary: last inverse pass in registers (stages RB .. 1, unit stride: the twiddles are
  // the same for every group), rescale, optionally publish the coefficients ----
  Fp keep[ITERS][NC][EB];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int g = tid + it * kLdeThreads;
    const bool act = NGB >= kLdeThreads || g < NGB;
    const int base = g << RB;
#pragma unroll
    for (int k = 0; k < EB; ++k)
#pragma unroll
      for (int c = 0; c < NC; ++c)
        keep[it][c][k] = (act && (FULL || c < nc)) ? buf[c * PADDED + lds_idx(base + k)] : Fp::zero();
#pragma unroll
    for (int st = 0; st < RB; ++st) {
      const int hk = EB >> (st + 1), stage = RB - st;
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        if ((k & hk) != 0) continue;
        const bool unit = (k & (hk - 1)) == 0;  // twiddle w^0 = 1 (every pair of stage 1): no multiplication
        const Fp w = unit ? Fp::one() : Fp::raw(tw_inv[(1 << (stage - 1)) + (k & (hk - 1))]);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const Fp u = keep[it][c][k], v = keep[it][c][k + hk];
          keep[it][c][k] = u + v;
          keep[it][c][k + hk] = unit ? u - v : (u - v) * w;
        }
      }
    }
    if (act) {
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        const Fp sc = Fp::raw(isc[base + k]);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          keep[it][c][k] = keep[it][c][k] * sc;
          if (cdst && (FULL || c < nc)) cdst[(size_t)c * geom.coef_col + base + k] = keep[it][c][k].v;
        }
      }
    }
  }
  __syncthreads();  // every thread has taken its coefficients out of the image
  // ---- two forward transforms: DIT stages 1 .. LOGH, the first pass from registers ----
#pragma unroll 1
  for (int cs = 0; cs < 2; ++cs) {
    const uint32_t* osc = out_scale_br + (size_t)cs * geom.coset;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int g = tid + it * kLdeThreads;
      const bool act = NGB >= kLdeThreads || g < NGB;
      const int base = g << RB;
      Fp y[NC][EB];
#pragma unroll
      for (int k = 0; k < EB; ++k) {
        const Fp sc = act ? Fp::raw(osc[base + k]) : Fp::zero();
#pragma unroll
        for (int c = 0; c < NC; ++c) y[c][k] = keep[it][c][k] * sc;
      }
#pragma unroll
      for (int st = 0; st < RB; ++st) {
        const int hk = 1 << st, stage = 1 + st;
#pragma unroll
        for (int k = 0; k < EB; ++k) {
          if ((k & hk) != 0) continue;
          const bool unit = (k & (hk - 1)) == 0;
          const Fp w = unit ? Fp::one() : Fp::raw(tw_fwd[(1 << (stage - 1)) + (k & (hk - 1))]);
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const Fp u = y[c][k], t = unit ? y[c][k + hk] : y[c][k + hk] * w;
            y[c][k] = u + t;
            y[c][k + hk] = u - t;
          }
        }
      }
      if (act) {
#pragma unroll
        for (int k = 0; k < EB; ++k)
#pragma unroll
          for (int c = 0; c < NC; ++c)
            if (FULL || c < nc) buf[c * PADDED + lds_idx(base + k)] = y[c][k];
      }
    }
    __syncthreads();
    io.src_glb = nullptr;
#pragma unroll
    for (int s = 1 + RB; s + 3 <= LOGH; s += 3) {
      io.dst_glb = nullptr;
      ntt_pass<3, false, NC, kDstLds, FULL>(io, tw_fwd, LOGH, s, nc, tid);
      __syncthreads();
    }
    io.dst_glb = dst + (size_t)cs * geom.coset;
    ntt_pass<3, false, NC, kDstGlb, FULL>(io, tw_fwd, LOGH, LOGH - 2, nc, tid);
    __syncthreads();  // the image is rewritten by the next coset / the next column group
  }
}

template <int LOGH, int NC>
__global__ __launch_bounds__(kLdeThreads) void lde_fixed_kernel(const uint32_t* __restrict__ in,
                                                                uint32_t* __restrict__ coefs_br,
                                                                uint32_t* __restrict__ out,
                                                                const uint32_t* __restrict__ tw_fwd,
                                                                const uint32_t* __restrict__ tw_inv,
                                                                const uint32_t* __restrict__ in_scale_br,
                                                                int scale_sel_shift, int scale_sel_mask,
                                                                const uint32_t* __restrict__ out_scale_br, size_t ncols) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  constexpr int H = 1 << LOGH;
  Fp* buf = reinterpret_cast<Fp*>(smem);  // NC images
  const int tid = threadIdx.x;
  const size_t ngroups = (ncols + NC - 1) / NC;
  for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const size_t col = grp * NC;
    const int nc = (int)((ncols - col) < (size_t)NC ? (ncols - col) : (size_t)NC);
    const uint32_t* src = in + col * (size_t)H;
    const uint32_t* isc = in_scale_br + (size_t)((col >> scale_sel_shift) & (size_t)scale_sel_mask) * H;
    uint32_t* cdst = coefs_br ? coefs_br + col * (size_t)H : nullptr;
    uint32_t* dst = out + col * 2 * (size_t)H;
    const LdeGeom geom{(size_t)H, (size_t)H, (size_t)2 * H, (size_t)H};
    if (nc == NC) lde_fixed_group<LOGH, NC, true>(src, cdst, dst, tw_fwd, tw_inv, isc, out_scale_br, buf, nc, tid, geom);
    else lde_fixed_group<LOGH, NC, false>(src, cdst, dst, tw_fwd, tw_inv, isc, out_scale_br, buf, nc, tid, geom);
  }
}

// ---------------------------------------------------------------------------
// Heights above 2^14 (the as-committed guest's 2^21-row chips, SURVEY.md section 8
// row f1): the column no longer fits LDS, so the H = H1*H2 transform is split the
// four-step way into two LDS-staged passes per direction:
//   "chunk" pass  : the low l2 stages, independent inside each contiguous run of
//                   H2 = 2^l2 elements (the same register passes as above);
//   "strided" pass: the high l1 stages, which couple elements H2 apart; a workgroup
//                   stages an H1 x 16 tile (16 adjacent n2, so every row is a 64-byte
//                   segment) and runs the l1 stages on it with the full-size twiddles.
// DIF runs strided then chunk (natural -> bit-reversed), DIT chunk then strided.
// ---------------------------------------------------------------------------
constexpr int kStrideTile = 16;

template <bool DIF>
__global__ __launch_bounds__(kLdeThreads) void ntt_chunk_kernel(const uint32_t* __restrict__ src,
                                                               uint32_t* __restrict__ dst, size_t src_col_stride,
                                                               size_t dst_col_stride,
                                                               const uint32_t* __restrict__ tw,
                                                               const uint32_t* __restrict__ pre_scale,
                                                               const uint32_t* __restrict__ post_scale, int logh,
                                                               int l2, int post_sel_shift, int post_sel_mask) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  Fp* buf = reinterpret_cast<Fp*>(smem);
  const size_t h2 = (size_t)1 << l2;
  const size_t chunk = blockIdx.x, col = blockIdx.y;
  const uint32_t* s0 = src + col * src_col_stride + chunk * h2;
  uint32_t* d0 = dst + col * dst_col_stride + chunk * h2;
  const uint32_t* pre = pre_scale ? pre_scale + chunk * h2 : nullptr;
  // the post-scale table may be chosen per column (quotient chunks were evaluated over different cosets)
  const uint32_t* post =
      post_scale ? post_scale + ((size_t)((col >> post_sel_shift) & (size_t)post_sel_mask) << logh) + chunk * h2 : nullptr;
  const int tid = threadIdx.x;
  int s = DIF ? l2 : 1;
  bool first = true;
  while (DIF ? s > 0 : s <= l2) {
    int r;
    bool last;
    if (DIF) {
      r = s >= 3 ? 3 : s;
      last = (s - r) == 0;
    } else {
      const int rem = l2 - s + 1;
      r = first ? ((rem % 3) ? (rem % 3) : 3) : 3;
      last = (s + r) > l2;
    }
    PassIo io;
    io.src_lds = buf;
    io.src_glb = first ? s0 : nullptr;
    io.pre_scale = first ? pre : nullptr;
    io.dst_lds = last ? nullptr : buf;
    io.dst_glb = last ? d0 : nullptr;
    io.post_scale = last ? post : nullptr;
    io.src_glb_stride = io.dst_glb_stride = 0;
    io.lds_stride = 0;
    ntt_pass_sel<DIF, 1, kDstLds, kSrcGlb | kDstLds, kSrcGlb | kPre | kDstLds, kDstGlb, kDstGlb | kPost, kSrcGlb | kDstGlb,
                 kSrcGlb | kDstGlb | kPost, kSrcGlb | kPre | kDstGlb, kSrcGlb | kPre | kDstGlb | kPost>(
        pass_flags(io), r, io, tw, l2, s, 1, tid);
    __syncthreads();
    s = DIF ? s - r : s + r;
    first = false;
  }
}

// the l1 = logh - l2 high stages of every column, from `src` to `dst` (which may be the same
// buffer: a tile is read completely before any of it is written back)
template <bool DIF>
__global__ __launch_bounds__(kLdeThreads) void ntt_strided_kernel(const uint32_t* src, size_t src_col_stride,
                                                                 uint32_t* dst, size_t dst_col_stride,
                                                                 const uint32_t* __restrict__ tw, int logh, int l2) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  Fp* buf = reinterpret_cast<Fp*>(smem);
  const int l1 = logh - l2;
  const int h1 = 1 << l1;
  const size_t h2 = (size_t)1 << l2;
  const size_t n2_base = (size_t)blockIdx.x * kStrideTile;
  const uint32_t* sp = src + (size_t)blockIdx.y * src_col_stride + n2_base;
  uint32_t* d = dst + (size_t)blockIdx.y * dst_col_stride + n2_base;
  const int tid = threadIdx.x;
  // LDS image [n1][t] with one pad word per row (17-word rows: conflict-free column walks)
  constexpr int kRow = kStrideTile + 1;
  for (int i = tid; i < h1 * kStrideTile; i += kLdeThreads) {
    const int n1 = i / kStrideTile, t = i % kStrideTile;
    buf[n1 * kRow + t] = Fp::raw(sp[(size_t)n1 * h2 + t]);
  }
  __syncthreads();
  const int nbf = (h1 >> 1) * kStrideTile;
  for (int st = 0; st < l1; ++st) {
    const int s1 = DIF ? l1 - st : st + 1;  // stage inside the H1 transform; full-size stage = l2 + s1
    const int half1 = 1 << (s1 - 1);
    const size_t tw_base = (size_t)1 << (l2 + s1 - 1);  // per-stage table of the full-size stage l2 + s1
    for (int i = tid; i < nbf; i += kLdeThreads) {
      const int t = i % kStrideTile, b = i / kStrideTile;
      const int j1 = b & (half1 - 1);
      const int lo = ((b >> (s1 - 1)) << s1) | j1, hi = lo + half1;
      // position inside the half block of the full-size stage: j1*H2 + n2
      const size_t j = ((size_t)j1 << l2) + n2_base + t;
      const Fp w = Fp::raw(tw[tw_base + j]);
      Fp u = buf[lo * kRow + t], v = buf[hi * kRow + t];
      if (DIF) {
        buf[lo * kRow + t] = u + v;
        buf[hi * kRow + t] = Fp::raw(fps_canon(fps_mul((int32_t)u.v - (int32_t)v.v, (int32_t)w.v)));  // (as in ntt_pass)
      } else {
        Fp x = v * w;
        buf[lo * kRow + t] = u + x;
        buf[hi * kRow + t] = u - x;
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < h1 * kStrideTile; i += kLdeThreads) {
    const int n1 = i / kStrideTile, t = i % kStrideTile;
    d[(size_t)n1 * h2 + t] = buf[n1 * kRow + t].v;
  }
}

// ---------------------------------------------------------------------------
// Heights 2^15 .. 2^21, second form (the machine proof's CPU chip lives here): H = 2^l1 * 2^l2.
//   ntt_top_kernel   the l1 stages that couple elements 2^l2 apart, in ONE register pass for l1 <= 6 (a
//                    thread owns the 2^l1 elements of one residue n2, stride 2^l2, so a wave's loads
//                    and stores are 64 consecutive words at every step; no LDS, no barrier); for
//                    2^20 and 2^21 (l1 = 7, 8) in two passes of 4 and l1 - 4 stages, since 128 elements
//                    per lane do not stay in registers
//   lde_chunk_kernel the l2 low stages of the inverse transform on one contiguous run of 2^l2
//                    elements in LDS, rescale, publish the coefficients, then the l2 low stages of
//                    BOTH forward transforms from the same LDS image
// Per column: inverse top (read H, write H), chunks (read H, write H coefficients + 2H), forward top
// in place (read 2H, write 2H): 10 column units instead of the 12 of the strided/chunk pairs above,
// every access a full line.
// ---------------------------------------------------------------------------
template <int R, bool DIF>
__global__ __launch_bounds__(kLdeThreads) void ntt_top_kernel(const uint32_t* src, size_t src_col_stride, uint32_t* dst,
                                                             size_t dst_col_stride, const uint32_t* __restrict__ tw,
                                                             int logh, int s) {
  PassIo io;
  io.src_lds = nullptr;
  io.dst_lds = nullptr;
  io.src_glb = src + (size_t)blockIdx.y * src_col_stride;
  io.dst_glb = dst + (size_t)blockIdx.y * dst_col_stride;
  io.pre_scale = io.post_scale = nullptr;
  io.src_glb_stride = io.dst_glb_stride = 0;
  io.lds_stride = 0;
  // DIF: stages s .. s - R + 1 (group stride 2^(s - R)); DIT: stages s .. s + R - 1 (group stride 2^(s - 1))
  ntt_pass<R, DIF, 1, kSrcGlb | kDstGlb, true>(io, tw, logh, s, 1,
                                              (int)(blockIdx.x * kLdeThreads + threadIdx.x), (int)(gridDim.x * kLdeThreads));
}

__global__ __launch_bounds__(kLdeThreads) void lde_chunk_kernel(const uint32_t* __restrict__ scratch, size_t scratch_col_stride,
                                                               uint32_t* __restrict__ coefs_br, uint32_t* __restrict__ out,
                                                               const uint32_t* __restrict__ tw_fwd,
                                                               const uint32_t* __restrict__ tw_inv,
                                                               const uint32_t* __restrict__ in_scale_br, int scale_sel_shift,
                                                               int scale_sel_mask, const uint32_t* __restrict__ out_scale_br,
                                                               int logh, int l2) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const size_t h = (size_t)1 << logh, h2 = (size_t)1 << l2;
  const int padded = (int)(h2 + (h2 >> 3) + 4);
  Fp* coef = reinterpret_cast<Fp*>(smem);
  Fp* work = coef + padded;
  const int tid = threadIdx.x;
  const size_t chunk = blockIdx.x, col = blockIdx.y, off = chunk * h2;
  const uint32_t* src = scratch + col * scratch_col_stride + off;
  const uint32_t* isc = in_scale_br + (((col >> scale_sel_shift) & (size_t)scale_sel_mask) << logh) + off;
  uint32_t* cdst = coefs_br + col * h + off;
  {
    int s = l2;
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
      io.src_glb_stride = io.dst_glb_stride = 0;
      io.lds_stride = padded;
      ntt_pass_sel<true, 1, kDstLds, kSrcGlb | kDstLds, kDstLds | kDstGlb | kPost, kSrcGlb | kDstLds | kDstGlb | kPost>(
          pass_flags(io), r, io, tw_inv, l2, s, 1, tid);
      __syncthreads();
      s -= r;
      first = false;
    }
  }
  for (int cs = 0; cs < 2; ++cs) {
    const uint32_t* osc = out_scale_br + (size_t)cs * h + off;
    uint32_t* dst = out + (col * 2 + cs) * h + off;
    int s = 1;
    bool first = true;
    while (s <= l2) {
      const int rem = l2 - s + 1;
      const int r = first ? ((rem % 3) ? (rem % 3) : 3) : 3;
      const bool last = (s + r) > l2;
      PassIo io;
      io.src_lds = first ? coef : work;
      io.src_glb = nullptr;
      io.pre_scale = first ? osc : nullptr;
      io.dst_lds = last ? nullptr : work;
      io.dst_glb = last ? dst : nullptr;
      io.post_scale = nullptr;
      io.src_glb_stride = io.dst_glb_stride = 0;
      io.lds_stride = padded;
      ntt_pass_sel<false, 1, kDstLds, kPre | kDstLds, kDstGlb, kPre | kDstGlb>(pass_flags(io), r, io, tw_fwd, l2, s, 1, tid);
      __syncthreads();
      s += r;
      first = false;
    }
  }
}

// The same chunk work for l2 = 11 .. 13 with the chunk length a template constant: one LDS image, the
// boundary pass in registers (lde_fixed_group), so twice the workgroups per CU and one LDS round trip
// less per transform than lde_chunk_kernel.
template <int L2>
__global__ __launch_bounds__(kLdeThreads) void lde_chunk_fixed_kernel(const uint32_t* __restrict__ scratch, size_t scratch_col_stride,
                                                                     uint32_t* __restrict__ coefs_br, uint32_t* __restrict__ out,
                                                                     const uint32_t* __restrict__ tw_fwd,
                                                                     const uint32_t* __restrict__ tw_inv,
                                                                     const uint32_t* __restrict__ in_scale_br, int scale_sel_shift,
                                                                     int scale_sel_mask, const uint32_t* __restrict__ out_scale_br,
                                                                     int logh) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const size_t h = (size_t)1 << logh;
  const size_t col = blockIdx.y, off = (size_t)blockIdx.x << L2;
  const uint32_t* src = scratch + col * scratch_col_stride + off;
  const uint32_t* isc = in_scale_br + (((col >> scale_sel_shift) & (size_t)scale_sel_mask) << logh) + off;
  const LdeGeom geom{0, 0, 0, h};
  lde_fixed_group<L2, 1, true>(src, coefs_br ? coefs_br + col * h + off : nullptr, out + col * 2 * h + off, tw_fwd, tw_inv, isc, out_scale_br + off,
                               reinterpret_cast<Fp*>(smem), 1, threadIdx.x, geom);
}

template <bool DIF>
static void launch_ntt_top(hipStream_t stream, int r, int s, const uint32_t* src, size_t src_stride, uint32_t* dst,
                           size_t dst_stride, const uint32_t* tw, int logh, size_t ncols) {
  const unsigned groups = (unsigned)(((size_t)1 << (logh - r)) / kLdeThreads);
  const dim3 grid(groups ? groups : 1, (unsigned)ncols), block(kLdeThreads);
  switch (r) {
    case 1: hipLaunchKernelGGL((ntt_top_kernel<1, DIF>), grid, block, 0, stream, src, src_stride, dst, dst_stride, tw, logh, s); break;
    case 2: hipLaunchKernelGGL((ntt_top_kernel<2, DIF>), grid, block, 0, stream, src, src_stride, dst, dst_stride, tw, logh, s); break;
    case 3: hipLaunchKernelGGL((ntt_top_kernel<3, DIF>), grid, block, 0, stream, src, src_stride, dst, dst_stride, tw, logh, s); break;
    case 4: hipLaunchKernelGGL((ntt_top_kernel<4, DIF>), grid, block, 0, stream, src, src_stride, dst, dst_stride, tw, logh, s); break;
    case 5: hipLaunchKernelGGL((ntt_top_kernel<5, DIF>), grid, block, 0, stream, src, src_stride, dst, dst_stride, tw, logh, s); break;
    case 6: hipLaunchKernelGGL((ntt_top_kernel<6, DIF>), grid, block, 0, stream, src, src_stride, dst, dst_stride, tw, logh, s); break;
    default: break;  // launch_lde_tall never asks for more than 6 stages in one pass
  }
}

static void launch_lde_tall(hipStream_t stream, const uint32_t* in, uint32_t* coefs_br, uint32_t* out, const uint32_t* tw_fwd,
                            const uint32_t* tw_inv, const uint32_t* in_scale_br, int scale_sel_shift, int scale_sel_mask,
                            const uint32_t* out_scale_br, int logh, size_t ncols) {
  const size_t h = (size_t)1 << logh;
  static const bool generic_chunk = getenv("ZKSP_LDE_GENERIC_CHUNK") != nullptr;  // debugging switch
  const bool fixed = !generic_chunk || !coefs_br;  // (the generic chunk kernel always writes the coefficients)
  // 2^13-point chunks; the l1 = logh - 13 strided stages in one register pass of at most 6 stages, or in two
  // (2^20, 2^21: a pass of 7 stages would hold 128 elements per lane)
  const int l2 = fixed ? 13 : (logh <= 19 ? 13 : 14), l1 = logh - l2;
  const int r_hi = l1 <= 6 ? l1 : 4, r_lo = l1 - r_hi;
  const size_t h2 = (size_t)1 << l2;
  const size_t smem = (fixed ? 1 : 2) * sizeof(uint32_t) * (h2 + (h2 >> 3) + 4);
  uint32_t* scratch = out + h;  // out[col][1]: every chunk of it is read, then rewritten, by the same workgroup
  // Columns in slabs: grid.y is limited to 65535, and a slab whose working set (input, scratch / LDE, coefficients: 16 H
  // bytes per column) fits the 256 MB Infinity Cache lets the four passes of a column hand their intermediates over
  // on-die instead of through HBM (ZKSP_LDE_SLAB_MB: the working set of a slab in MB; default 0 = no blocking: measured on MI355X at 2^16 and 2^18, slabs of 96-224 MB ran 2-25 % SLOWER than one launch over all columns - the passes are bound by butterfly issue and LDS, not by HBM).
  static const size_t slab_mb = getenv("ZKSP_LDE_SLAB_MB") ? (size_t)atoi(getenv("ZKSP_LDE_SLAB_MB")) : 0;
  size_t slab = 16384;
  // (a multiple of 8 columns: the chunk kernel picks the input scale table from the column index inside the slab, and
  // quotient chunks alternate tables every 4 columns)
  if (slab_mb) slab = std::min<size_t>(slab, std::max<size_t>(8, ((slab_mb << 20) / (16 * h)) & ~(size_t)7));
  for (size_t c0 = 0; c0 < ncols; c0 += slab) {
    const size_t nc = ncols - c0 < slab ? ncols - c0 : slab;
    if (r_hi) launch_ntt_top<true>(stream, r_hi, logh, in + c0 * h, h, scratch + c0 * 2 * h, 2 * h, tw_inv, logh, nc);
    if (r_lo) launch_ntt_top<true>(stream, r_lo, logh - r_hi, scratch + c0 * 2 * h, 2 * h, scratch + c0 * 2 * h, 2 * h, tw_inv, logh, nc);
    if (fixed)  // (a 2^13 column is one chunk: no top passes, the chunk kernel reads the input itself)
      hipLaunchKernelGGL(lde_chunk_fixed_kernel<13>, dim3((unsigned)(h >> l2), (unsigned)nc), dim3(kLdeThreads), smem, stream,
                         l1 ? scratch + c0 * 2 * h : in + c0 * h, l1 ? 2 * h : h, coefs_br ? coefs_br + c0 * h : nullptr, out + c0 * 2 * h,
                         tw_fwd, tw_inv, in_scale_br, scale_sel_shift, scale_sel_mask, out_scale_br, logh);
    else
      hipLaunchKernelGGL(lde_chunk_kernel, dim3((unsigned)(h >> l2), (unsigned)nc), dim3(kLdeThreads), smem, stream,
                         scratch + c0 * 2 * h, 2 * h, coefs_br + c0 * h, out + c0 * 2 * h, tw_fwd, tw_inv, in_scale_br,
                         scale_sel_shift, scale_sel_mask, out_scale_br, logh, l2);
    // forward top stages in place, both cosets: (column, coset) pairs are h words apart
    if (r_lo) launch_ntt_top<false>(stream, r_lo, l2 + 1, out + c0 * 2 * h, h, out + c0 * 2 * h, h, tw_fwd, logh, 2 * nc);
    if (r_hi) launch_ntt_top<false>(stream, r_hi, l2 + 1 + r_lo, out + c0 * 2 * h, h, out + c0 * 2 * h, h, tw_fwd, logh, 2 * nc);
  }
}

static void launch_lde_large(hipStream_t stream, const uint32_t* in, uint32_t* coefs_br, uint32_t* out,
                             const uint32_t* tw_fwd, const uint32_t* tw_inv, const uint32_t* in_scale_br,
                             int scale_sel_shift, int scale_sel_mask, const uint32_t* out_scale_br, int logh, size_t ncols) {
  const size_t h = (size_t)1 << logh;
  const int l2 = logh - logh / 2;  // low (contiguous) stages: 2^l2 <= 2^11 per chunk for logh <= 22
  const int l1 = logh - l2;        // high (strided) stages: an H1 x 16 tile is at most 2^11 x 17 words of LDS
  const size_t chunk_smem = sizeof(uint32_t) * (((size_t)1 << l2) + ((size_t)1 << l2 >> 3) + 4);
  const size_t strided_smem = sizeof(uint32_t) * ((size_t)1 << l1) * (kStrideTile + 1);
  const dim3 chunk_grid((unsigned)(h >> l2), (unsigned)ncols);
  const dim3 sgrid((unsigned)(((size_t)1 << l2) / kStrideTile), (unsigned)ncols);
  // inverse: strided DIF from the input into the second coset slot of the output (scratch), chunk DIF -> coefs
  uint32_t* scratch = out + h;  // out[col][1]
  hipLaunchKernelGGL(ntt_strided_kernel<true>, sgrid, dim3(kLdeThreads), strided_smem, stream, in, h, scratch, 2 * h,
                     tw_inv, logh, l2);
  hipLaunchKernelGGL(ntt_chunk_kernel<true>, chunk_grid, dim3(kLdeThreads), chunk_smem, stream, scratch, coefs_br,
                     2 * h, h, tw_inv, (const uint32_t*)nullptr, in_scale_br, logh, l2, scale_sel_shift, scale_sel_mask);
  for (int cs = 0; cs < 2; ++cs) {
    uint32_t* dst = out + (size_t)cs * h;
    hipLaunchKernelGGL(ntt_chunk_kernel<false>, chunk_grid, dim3(kLdeThreads), chunk_smem, stream, coefs_br, dst, h,
                       2 * h, tw_fwd, out_scale_br + (size_t)cs * h, (const uint32_t*)nullptr, logh, l2, 0, 0);
    hipLaunchKernelGGL(ntt_strided_kernel<false>, sgrid, dim3(kLdeThreads), strided_smem, stream, dst, 2 * h, dst, 2 * h,
                       tw_fwd, logh, l2);
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
  static const bool mid_generic = getenv("ZKSP_LDE_MID_GENERIC") != nullptr;  // debugging switch: 2^13, 2^14 through lde_lds_kernel
  if (logh > 14 || (logh >= 13 && !coefs_br && !mid_generic)) {
    static const bool old_tall = getenv("ZKSP_LDE_OLD_TALL") != nullptr;  // debugging switch: the strided/chunk form
    if (logh <= 21 && (!coefs_br || !old_tall))
      launch_lde_tall(stream, in, coefs_br, out, tw_fwd, tw_inv, in_scale_br, scale_sel_shift, scale_sel_mask, out_scale_br, logh,
                      ncols);
    else
      launch_lde_large(stream, in, coefs_br, out, tw_fwd, tw_inv, in_scale_br, scale_sel_shift, scale_sel_mask, out_scale_br,
                       logh, ncols);
    return;
  }
  const size_t h = (size_t)1 << logh;
  const int nc = lde_cols_per_block(logh);
  const size_t groups = (ncols + nc - 1) / nc;
  const unsigned grid = (unsigned)(groups < 65536 ? groups : 65536);
  static const bool fixed_off = getenv("ZKSP_LDE_GENERIC") != nullptr;  // debugging switch
  if (nc == 2 && logh >= 9 && logh <= 11 && !fixed_off) {
    const size_t smem1 = (size_t)2 * sizeof(uint32_t) * (h + (h >> 3) + 4);  // one image per column
    if (logh == 11)
      hipLaunchKernelGGL((lde_fixed_kernel<11, 2>), dim3(grid), dim3(kLdeThreads), smem1, stream, in, coefs_br, out, tw_fwd,
                         tw_inv, in_scale_br, scale_sel_shift, scale_sel_mask, out_scale_br, ncols);
    else if (logh == 10)
      hipLaunchKernelGGL((lde_fixed_kernel<10, 2>), dim3(grid), dim3(kLdeThreads), smem1, stream, in, coefs_br, out, tw_fwd,
                         tw_inv, in_scale_br, scale_sel_shift, scale_sel_mask, out_scale_br, ncols);
    else
      hipLaunchKernelGGL((lde_fixed_kernel<9, 2>), dim3(grid), dim3(kLdeThreads), smem1, stream, in, coefs_br, out, tw_fwd,
                         tw_inv, in_scale_br, scale_sel_shift, scale_sel_mask, out_scale_br, ncols);
    return;
  }
  const size_t smem = (size_t)nc * 2 * sizeof(uint32_t) * (h + (h >> 3) + 4);
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
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(lde_chunk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
  const void* big[] = {reinterpret_cast<const void*>(ntt_chunk_kernel<true>),
                       reinterpret_cast<const void*>(ntt_chunk_kernel<false>),
                       reinterpret_cast<const void*>(ntt_strided_kernel<true>),
                       reinterpret_cast<const void*>(ntt_strided_kernel<false>)};
  for (const void* f : big) {
    e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(lde_lds_kernel<2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace zksp
