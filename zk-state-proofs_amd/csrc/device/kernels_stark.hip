// Device side of the STARK prover around the NTT and Merkle kernels:
// keccak trace generation (row a3), constraint/quotient evaluation (row a6),
// out-of-domain openings, reduced openings and FRI folding (row a7), the
// duplex-sponge Fiat-Shamir transcript (row a8, prover side) and proof assembly.
// Replaces what sp1-core-machine / p3-uni-stark / p3-fri / p3-challenger do on the
// host beneath the reference's `client.prove(&pk, stdin).run()`
// (prover/src/bin/main.rs:71-74; Cargo.lock:7130, :5378, :5253, :5197).
//
// Batch convention: the leading (slowest) index of every buffer is the proof in
// the batch; kernels take it from blockIdx.y (or one lane per proof for the
// transcript kernels, so that a whole batch of sponges advances in one wave).
#include "air_keccak.hpp"
#include "kernels.h"
#include "merkle_coop.hpp"

#include <algorithm>
#include <cstdlib>

namespace zksp {

constexpr int kThreads = 256;

__device__ __forceinline__ Fp4 load_fp4(const uint32_t* p) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  Fp4 r;
  r.c[0] = Fp::raw(v.x); r.c[1] = Fp::raw(v.y); r.c[2] = Fp::raw(v.z); r.c[3] = Fp::raw(v.w);
  return r;
}
__device__ __forceinline__ void store_fp4(uint32_t* p, const Fp4& a) {
  *reinterpret_cast<uint4*>(p) = make_uint4(a.c[0].v, a.c[1].v, a.c[2].v, a.c[3].v);
}

// ===========================================================================
// keccak trace generation: one lane per trace row
// ===========================================================================
__device__ __forceinline__ uint64_t rol64(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }

__global__ __launch_bounds__(kThreads) void keccak_trace_kernel(const uint64_t* __restrict__ states, int max_perms,
                                                               const uint32_t* __restrict__ n_perms,
                                                               uint32_t* __restrict__ trace, size_t trace_bstride, int logh) {
  const int h = 1 << logh;
  const int row = blockIdx.x * kThreads + threadIdx.x;
  if (row >= h) return;
  const int b = blockIdx.y;
  const int perm = row / 24, round = row - perm * 24;
  const int np = (int)n_perms[b];
  const ka::Tables& T = ka::tables();
  uint64_t pre[25], a[25];
  if (perm < np) {
    const uint64_t* s = states + ((size_t)b * max_perms + perm) * 25;
#pragma unroll
    for (int i = 0; i < 25; ++i) pre[i] = s[i];
  } else {
#pragma unroll
    for (int i = 0; i < 25; ++i) pre[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < 25; ++i) a[i] = pre[i];
  uint64_t c[5], cp[5], ap[25], bb[25], app[25];
  // advance to this row's round; the last iteration's intermediates are the row
  for (int r = 0;; ++r) {
#pragma unroll
    for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
    for (int x = 0; x < 5; ++x) cp[x] = c[x] ^ c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
#pragma unroll
    for (int j = 0; j < 25; ++j) ap[j] = a[j] ^ c[j % 5] ^ cp[j % 5];
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
      for (int y = 0; y < 5; ++y) bb[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(ap[x + 5 * y], T.rot[x][y]);
#pragma unroll
    for (int y = 0; y < 5; ++y)
#pragma unroll
      for (int x = 0; x < 5; ++x) app[x + 5 * y] = bb[x + 5 * y] ^ (~bb[(x + 1) % 5 + 5 * y] & bb[(x + 2) % 5 + 5 * y]);
    if (r == round) break;
#pragma unroll
    for (int j = 0; j < 25; ++j) a[j] = app[j];
    a[0] ^= T.rc[r];
  }
  const uint64_t appp00 = app[0] ^ T.rc[round];
  uint32_t* t = trace + (size_t)b * trace_bstride + row;
  const size_t cs = (size_t)h;
  const uint32_t one = kR1;
  for (int i = 0; i < 24; ++i) t[(ka::kFlags + i) * cs] = (i == round) ? one : 0u;
  t[ka::kExport * cs] = (perm < np && round == 23) ? one : 0u;
#pragma unroll
  for (int j = 0; j < 25; ++j)
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      t[(ka::kPreimage + 4 * j + l) * cs] = Fp::from_canonical((uint32_t)((pre[j] >> (16 * l)) & 0xffff)).v;
      t[(ka::kA + 4 * j + l) * cs] = Fp::from_canonical((uint32_t)((a[j] >> (16 * l)) & 0xffff)).v;
      t[(ka::kApp + 4 * j + l) * cs] = Fp::from_canonical((uint32_t)((app[j] >> (16 * l)) & 0xffff)).v;
    }
#pragma unroll
  for (int x = 0; x < 5; ++x)
    for (int z = 0; z < 64; ++z) {
      t[(ka::kC + 64 * x + z) * cs] = ((c[x] >> z) & 1) ? one : 0u;
      t[(ka::kCp + 64 * x + z) * cs] = ((cp[x] >> z) & 1) ? one : 0u;
    }
#pragma unroll
  for (int j = 0; j < 25; ++j)
    for (int z = 0; z < 64; ++z) t[(ka::kAp + 64 * j + z) * cs] = ((ap[j] >> z) & 1) ? one : 0u;
  for (int z = 0; z < 64; ++z) t[(ka::kApp00 + z) * cs] = ((app[0] >> z) & 1) ? one : 0u;
#pragma unroll
  for (int l = 0; l < 4; ++l)
    t[(ka::kAppp00 + l) * cs] = Fp::from_canonical((uint32_t)((appp00 >> (16 * l)) & 0xffff)).v;
}

#ifdef ZKSP_COMPONENT  // the round-1 keccak-chip component path only (include/zksp_component.h)
void launch_keccak_trace(hipStream_t stream, const uint64_t* states, int max_perms, const uint32_t* n_perms,
                         uint32_t* trace, int logh, int batch) {
  const int h = 1 << logh;
  hipLaunchKernelGGL(keccak_trace_kernel, dim3((h + kThreads - 1) / kThreads, batch), dim3(kThreads), 0, stream,
                     states, max_perms, n_perms, trace, (size_t)ka::kWidth * h, logh);
}
#endif  // ZKSP_COMPONENT
// the same columns inside a wider per-proof matrix (the machine proof's keccak chip adds a column)
void launch_keccak_trace_strided(hipStream_t stream, const uint64_t* states, int max_perms, const uint32_t* n_perms,
                                 uint32_t* trace, size_t trace_bstride, int logh, int batch) {
  const int h = 1 << logh;
  hipLaunchKernelGGL(keccak_trace_kernel, dim3((h + kThreads - 1) / kThreads, batch), dim3(kThreads), 0, stream,
                     states, max_perms, n_perms, trace, trace_bstride, logh);
}

#ifdef ZKSP_COMPONENT  // the round-1 keccak-chip component path only (include/zksp_component.h)
// ===========================================================================
// constraint evaluation on the LDE domain -> quotient values
// ===========================================================================
struct QuotCtx {
  using F = Fp;
  using E = Fp4;
  const uint32_t* loc;
  const uint32_t* nxt;
  size_t cs;
  Fp first, trans, last;
  const uint32_t* ap;  // this proof's alpha powers (Fp4 each), indexed by constraint
  Fp4 acc;             // extension-valued constraints, and the folded total at the end
  int64_t lazy[4];     // signed lazy sum of alpha^k_i * c_k per coordinate (field.hpp)
  int pending;
  // LogUp bus
  const uint32_t* ploc;  // running-sum columns at this point / the next row (stride cs)
  const uint32_t* pnxt;
  const uint32_t* bus_ch;     // gamma, beta
  const uint32_t* beta_pows;  // [200] Fp4
  const uint32_t* cum;        // Fp4
  __device__ __forceinline__ F local(int col) const { return Fp::raw(loc[(size_t)col * cs]); }
  __device__ __forceinline__ F next(int col) const { return Fp::raw(nxt[(size_t)col * cs]); }
  __device__ __forceinline__ F is_first() const { return first; }
  __device__ __forceinline__ F is_trans() const { return trans; }
  __device__ __forceinline__ F is_last() const { return last; }
  __device__ __forceinline__ F one() const { return Fp::one(); }
  // acc += alpha^k * v, coordinate by coordinate: one multiply-add each into a signed 64-bit
  // accumulator.  alpha^k is uniform, so centring it is scalar work; v is canonical, so
  // |term| < p^2 / 2 and four terms fit between shrinks (a shrink keeps the residue and
  // brings the accumulator below 2^59).  One reduction per coordinate at the very end.
  __device__ __forceinline__ void emit_at(int k, F v) {
    const uint32_t* p = ap + 4 * (size_t)k;  // wave-uniform address: scalar loads
#pragma unroll
    for (int i = 0; i < 4; ++i) lazy[i] += (int64_t)fps_centre(p[i]) * (int64_t)v.v;
    if (++pending == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) lazy[i] = (int64_t)fps_fold(lazy[i]) * (int64_t)kRModP;
      pending = 0;
    }
  }
  __device__ __forceinline__ void flush() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc.c[i] = acc.c[i] + Fp::raw(fps_canon(fps_fold(lazy[i])));
      lazy[i] = 0;
    }
    pending = 0;
  }
  // extension-valued constraints of the bus
  __device__ __forceinline__ E gamma() const { return load_fp4(bus_ch); }
  __device__ __forceinline__ E beta_pow(int j) const { return load_fp4(beta_pows + 4 * (size_t)j); }
  __device__ __forceinline__ E cum_sum() const { return load_fp4(cum); }
  __device__ __forceinline__ E phi_local() const {
    Fp4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r.c[j] = Fp::raw(ploc[(size_t)j * cs]);
    return r;
  }
  __device__ __forceinline__ E phi_next() const {
    Fp4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r.c[j] = Fp::raw(pnxt[(size_t)j * cs]);
    return r;
  }
  __device__ __forceinline__ E lift(F v) const { return Fp4::from_base(v); }
  __device__ __forceinline__ void emit_ext_at(int k, E v) { acc += load_fp4(ap + 4 * (size_t)k) * v; }
};

__global__ __launch_bounds__(kThreads) void keccak_quotient_kernel(QuotientArgs a, int tiles_per_proof, int total_tiles) {
  const int h = 1 << a.logh, n = 2 * h;
  // A tile = 256 consecutive LDE points of one proof; its evaluation tasks read
  // overlapping column sets.  Workgroups are dealt round-robin over the 8 XCDs, so
  // give every XCD whole tiles: all tasks of a tile run back to back behind ONE L2
  // and each column slice leaves HBM once.  (Placement only affects speed; any
  // mapping is correct.)
  int tile, g;
  if ((total_tiles & 7) == 0) {
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    tile = (seq / ka::kNumTasks) * 8 + xcd;
    g = seq % ka::kNumTasks;
  } else {
    tile = blockIdx.x / ka::kNumTasks;
    g = blockIdx.x % ka::kNumTasks;
  }
  const int b = tile / tiles_per_proof;
  const int pt = (tile - b * tiles_per_proof) * kThreads + threadIdx.x;
  if (pt >= n) return;
  const int c = pt >= h ? 1 : 0, m = pt - c * h;
  const int mn = (m + 1) & (h - 1);
  const uint32_t* base = a.lde + (size_t)b * ka::kWidth * n + (size_t)c * h;
  const uint32_t* pbase = a.lde_p + (size_t)b * ka::kPermWidth * n + (size_t)c * h;
  QuotCtx ctx;
  ctx.loc = base + m;
  ctx.nxt = base + mn;
  ctx.cs = (size_t)n;
  ctx.first = Fp::raw(a.sel_first[pt]);
  ctx.trans = Fp::raw(a.sel_trans[pt]);
  ctx.last = Fp::raw(a.sel_last[pt]);
  ctx.ap = a.alpha_pows + (size_t)b * ka::kNumAllConstraints * 4;
  ctx.ploc = pbase + m;
  ctx.pnxt = pbase + mn;
  ctx.bus_ch = a.bus_ch + (size_t)b * 8;
  ctx.beta_pows = a.beta_pows + (size_t)b * ka::kBusTuple * 4;
  ctx.cum = a.cum_sum + (size_t)b * 4;
  ctx.acc = Fp4::zero();
  ctx.lazy[0] = ctx.lazy[1] = ctx.lazy[2] = ctx.lazy[3] = 0;
  ctx.pending = 0;
  if (g == ka::kBusTask) ka::eval_bus(ctx);
  else ka::eval_task(g, ctx);
  ctx.flush();
  store_fp4(a.partial + (((size_t)b * ka::kNumTasks + g) * n + pt) * 4, ctx.acc);
}

__global__ __launch_bounds__(kThreads) void keccak_quotient_combine_kernel(const uint32_t* __restrict__ partial,
                                                                          const uint32_t* __restrict__ zh_inv,
                                                                          uint32_t* __restrict__ quot, int logh) {
  const int h = 1 << logh, n = 2 * h;
  const int pt = blockIdx.x * kThreads + threadIdx.x;
  if (pt >= n) return;
  const int b = blockIdx.y;
  const int c = pt >= h ? 1 : 0, m = pt - c * h;
  Fp4 acc = Fp4::zero();
  for (int g = 0; g < ka::kNumTasks; ++g) acc += load_fp4(partial + (((size_t)b * ka::kNumTasks + g) * n + pt) * 4);
  acc = acc * Fp::raw(zh_inv[c]);
  uint32_t* q = quot + (size_t)b * 8 * h + m;
#pragma unroll
  for (int j = 0; j < 4; ++j) q[(size_t)(4 * c + j) * h] = acc.c[j].v;
}

void launch_keccak_quotient(hipStream_t stream, const QuotientArgs& a) {
  const int n = 2 << a.logh;
  const int blocks = (n + kThreads - 1) / kThreads;
  const int total_tiles = blocks * a.batch;
  hipLaunchKernelGGL(keccak_quotient_kernel, dim3((unsigned)total_tiles * ka::kNumTasks), dim3(kThreads), 0, stream, a,
                     blocks, total_tiles);
  hipLaunchKernelGGL(keccak_quotient_combine_kernel, dim3(blocks, a.batch), dim3(kThreads), 0, stream, a.partial,
                     a.zh_inv, a.quot, a.logh);
}

#endif  // ZKSP_COMPONENT

// ===========================================================================
// powers of an extension element (optionally stored in bit-reversed order)
// ===========================================================================
__global__ __launch_bounds__(kThreads) void ext_powers_kernel(const uint32_t* __restrict__ base, size_t base_stride,
                                                             uint32_t base_mul, uint32_t* __restrict__ out,
                                                             size_t out_stride, int n, int bitrev_logn, int centred) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const int b = blockIdx.y;
  uint32_t e = (uint32_t)i;
  if (bitrev_logn > 0) e = __brev(e) >> (32 - bitrev_logn);
  Fp4 x = load_fp4(base + (size_t)b * base_stride) * Fp::raw(base_mul);
  Fp4 r = Fp4::one();
  while (e) {
    if (e & 1) r = r * x;
    x = x.sqr();
    e >>= 1;
  }
  if (centred) {
    // signed words in (-p/2, p/2] for consumers that accumulate lazily (open_kernel)
#pragma unroll
    for (int j = 0; j < 4; ++j) r.c[j] = Fp::raw((uint32_t)fps_centre(r.c[j].v));
  }
  store_fp4(out + (size_t)b * out_stride + (size_t)i * 4, r);
}

// The bit-reversed table for n = 2^logn >= 64: a lane writes 16 consecutive entries i = 16 t + j, whose exponents are
// brev(i) = brev_4(j) * 2^(logn-4) + brev_(logn-4)(t).  One square-and-multiply chain gives both x^brev(t) and
// y = x^(2^(logn-4)); the 16 entries are x^brev(t) * y^k: about 3 extension products per entry instead of 1.5 logn.
__global__ __launch_bounds__(kThreads) void ext_powers_brev_kernel(const uint32_t* __restrict__ base, size_t base_stride,
                                                                  uint32_t base_mul, uint32_t* __restrict__ out,
                                                                  size_t out_stride, int logn, int centred) {
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= (1 << (logn - 4))) return;
  const int b = blockIdx.y;
  Fp4 x = load_fp4(base + (size_t)b * base_stride) * Fp::raw(base_mul);
  Fp4 r = Fp4::one();
  uint32_t e = __brev((uint32_t)t) >> (32 - (logn - 4));
  for (int k = 0; k < logn - 4; ++k) {
    if (e & 1) r = r * x;
    x = x.sqr();
    e >>= 1;
  }
  Fp4 yk = Fp4::one();  // y^k, k = 0..15; entry j takes k = brev_4(j)
  uint32_t* o = out + (size_t)b * out_stride + (size_t)t * 64;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    Fp4 v = r * yk;
    if (centred) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v.c[j] = Fp::raw((uint32_t)fps_centre(v.c[j].v));
    }
    const int j = ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3);
    store_fp4(o + 4 * j, v);
    yk = yk * x;
  }
}

void launch_ext_powers(hipStream_t stream, const uint32_t* base, size_t base_stride, uint32_t base_mul, uint32_t* out,
                       size_t out_stride, int n, int bitrev_logn, int batch, int centred) {
  if (bitrev_logn >= 6 && n == (1 << bitrev_logn)) {
    const int lanes = n >> 4;
    hipLaunchKernelGGL(ext_powers_brev_kernel, dim3((lanes + kThreads - 1) / kThreads, batch), dim3(kThreads), 0, stream, base,
                       base_stride, base_mul, out, out_stride, bitrev_logn, centred);
    return;
  }
  hipLaunchKernelGGL(ext_powers_kernel, dim3((n + kThreads - 1) / kThreads, batch), dim3(kThreads), 0, stream, base,
                     base_stride, base_mul, out, out_stride, n, bitrev_logn, centred);
}

// ===========================================================================
// barycentric weights: opening a column from its EVALUATIONS v_i = p(sigma w^i) on a coset of the height-H subgroup,
//   p(z) = ((y^H - 1) / H) * sum_i v_i w^i / (y - w^i),   y = z / sigma,
// is the same dot product the opening kernels compute against a table of powers, with the table
//   W_i = ((y^H - 1) / H) * w^i / (y - w^i)
// instead (so no coefficient array has to be kept).  The table for z * w is the same table moved by one place,
// W'_i = W_{i-1}: written here as a second copy.  Entries are centred signed words like the power tables'.
// A lane inverts kBaryK denominators with one extension-field inversion (Montgomery's trick).
// ===========================================================================
constexpr int kBaryK = 16;
struct BarySigma { uint32_t inv[3]; };
// tables [4][H] per proof: 0 = the subgroup itself (sigma = 1), 1 = table 0 moved by one place, 2 and 3 = the cosets
// sigma.inv[1], sigma.inv[2] (given as 1 / sigma).  blockIdx.z picks the coset; a lane's kBaryK entries are kThreads
// apart, so that adjacent lanes read and write adjacent entries.
__global__ __launch_bounds__(kThreads) void bary_weights_kernel(const uint32_t* __restrict__ zeta, size_t zeta_stride, BarySigma sigma,
                                                               const uint32_t* __restrict__ tw_fwd, uint32_t h_inv,
                                                               uint32_t* __restrict__ out, size_t out_stride, int logh) {
  const int h = 1 << logh, half = h >> 1;
  const int per_block = kThreads * kBaryK;
  const int i0 = blockIdx.x * per_block + threadIdx.x;  // entries i0 + k * step
  const int step = h < per_block ? h / kBaryK : kThreads;
  if (threadIdx.x >= step) return;
  const int b = blockIdx.y, which = blockIdx.z;
  const uint32_t sinv = which == 0 ? sigma.inv[0] : which == 1 ? sigma.inv[1] : sigma.inv[2];
  const Fp4 y = load_fp4(zeta + (size_t)b * zeta_stride) * Fp::raw(sinv);
  Fp4 yn = y;
  for (int k = 0; k < logh; ++k) yn = yn.sqr();
  const Fp4 scal = (yn - Fp4::one()) * Fp::raw(h_inv);
  Fp w[kBaryK];
  Fp4 pre[kBaryK];
#pragma unroll
  for (int k = 0; k < kBaryK; ++k) {
    const int i = i0 + k * step;
    w[k] = i < half ? Fp::raw(tw_fwd[i]) : -Fp::raw(tw_fwd[i - half]);
    Fp4 d = y;
    d.c[0] -= w[k];
    pre[k] = k ? pre[k - 1] * d : d;
  }
  Fp4 inv = pre[kBaryK - 1].inv() * scal;  // the common factor rides on the one inversion
  uint32_t* o = out + (size_t)b * out_stride + (size_t)(which == 0 ? 0 : which + 1) * h * 4;
#pragma unroll
  for (int k = kBaryK - 1; k >= 0; --k) {
    Fp4 v = (k ? inv * pre[k - 1] : inv) * w[k];
    Fp4 d = y;
    d.c[0] -= w[k];
    inv = inv * d;
#pragma unroll
    for (int j = 0; j < 4; ++j) v.c[j] = Fp::raw((uint32_t)fps_centre(v.c[j].v));
    const int i = i0 + k * step;
    store_fp4(o + (size_t)i * 4, v);
    if (which == 0) store_fp4(o + (size_t)h * 4 + (size_t)((i + 1) & (h - 1)) * 4, v);
  }
}
void launch_bary_weights(hipStream_t stream, const uint32_t* zeta, size_t zeta_stride, const uint32_t sigma_inv[3], const uint32_t* tw_fwd,
                         uint32_t h_inv, uint32_t* out, size_t out_stride, int logh, int batch) {
  const int h = 1 << logh, per_block = kThreads * kBaryK;
  BarySigma sg{{sigma_inv[0], sigma_inv[1], sigma_inv[2]}};
  hipLaunchKernelGGL(bary_weights_kernel, dim3((h + per_block - 1) / per_block, batch, 3), dim3(kThreads), 0, stream, zeta, zeta_stride,
                     sg, tw_fwd, h_inv, out, out_stride, logh);
}

// ===========================================================================
// out-of-domain openings: p(z) = sum_k coef_k z^k, one workgroup per column
// ===========================================================================
__device__ __forceinline__ Fp4 block_sum(Fp4 v, Fp4* red) {
  // wave reduction by shuffles, then across the 4 waves through LDS
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    Fp4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.c[j] = Fp::raw(__shfl_down(v.c[j].v, off, 64));
    v += o;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  Fp4 r = red[0];
  for (int w = 1; w < kThreads / 64; ++w) r += red[w];
  return r;
}

// A signed 64-bit accumulator brought back to a few bits above 2^58 without changing its
// residue: fold divides by 2^32, the multiplication by c = 2^32 mod p undoes that.
__device__ __forceinline__ int64_t lazy_shrink(int64_t t) { return (int64_t)fps_fold(t) * (int64_t)kRModP; }

// One workgroup opens kOpenCols columns, so a table of powers (64 KB for two points at
// H = 2^11) is read from L2 once per kOpenCols columns instead of once per column.  zpow holds
// CENTRED signed words (launch_ext_powers(.., centred = 1)); the coefficient is centred here,
// so |term| < p^2 / 4 and eight terms fit a signed 64-bit accumulator between shrinks.
constexpr int kOpenCols = 4;

__device__ __forceinline__ void open_body(const uint32_t* __restrict__ coefs, size_t coefs_stride, int ncols, int logh,
                                          const uint32_t* __restrict__ zpow, size_t zpow_stride, int npoints,
                                          uint32_t* __restrict__ opened, size_t opened_stride, size_t pt_stride, int bx, int b) {
  __shared__ Fp4 red[kThreads / 64];
  const int h = 1 << logh;
  const int col0 = bx * kOpenCols;
  const int nc = min(kOpenCols, ncols - col0);
  const uint32_t* cf = coefs + (size_t)b * coefs_stride + (size_t)col0 * h;
  const uint32_t* z0 = zpow + (size_t)b * zpow_stride;
  const uint32_t* z1 = z0 + (size_t)h * 4;
  int64_t acc[kOpenCols][2][4];
#pragma unroll
  for (int c = 0; c < kOpenCols; ++c)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[c][q][j] = 0;
  int pending = 0;
  for (int k = threadIdx.x; k < h; k += kThreads) {
    const uint4 p0 = *reinterpret_cast<const uint4*>(z0 + (size_t)k * 4);
    uint4 p1 = make_uint4(0, 0, 0, 0);
    if (npoints > 1) p1 = *reinterpret_cast<const uint4*>(z1 + (size_t)k * 4);
    const int32_t zz[2][4] = {{(int32_t)p0.x, (int32_t)p0.y, (int32_t)p0.z, (int32_t)p0.w},
                              {(int32_t)p1.x, (int32_t)p1.y, (int32_t)p1.z, (int32_t)p1.w}};
#pragma unroll
    for (int c = 0; c < kOpenCols; ++c) {
      if (c < nc) {
        const int32_t cv = fps_centre(cf[(size_t)c * h + k]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[c][0][j] += (int64_t)cv * (int64_t)zz[0][j];
          if (npoints > 1) acc[c][1][j] += (int64_t)cv * (int64_t)zz[1][j];
        }
      }
    }
    if (++pending == 8) {
#pragma unroll
      for (int c = 0; c < kOpenCols; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[c][q][j] = lazy_shrink(acc[c][q][j]);
      pending = 0;
    }
  }
#pragma unroll
  for (int c = 0; c < kOpenCols; ++c) {
    if (c >= nc) break;  // uniform: nc depends on blockIdx only
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q >= npoints) break;
      Fp4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v.c[j] = Fp::raw(fps_canon(fps_fold(acc[c][q][j])));
      const Fp4 sum = block_sum(v, red);
      if (threadIdx.x == 0)
        store_fp4(opened + (size_t)b * opened_stride + ((size_t)q * pt_stride + (size_t)(col0 + c)) * 4, sum);
    }
  }
}

__global__ __launch_bounds__(kThreads) void open_kernel(const uint32_t* __restrict__ coefs, size_t coefs_stride,
                                                       int ncols, int logh, const uint32_t* __restrict__ zpow,
                                                       size_t zpow_stride, int npoints, uint32_t* __restrict__ opened,
                                                       size_t opened_stride, size_t pt_stride) {
  open_body(coefs, coefs_stride, ncols, logh, zpow, zpow_stride, npoints, opened, opened_stride, pt_stride, blockIdx.x, blockIdx.y);
}

// Wide matrices (the 2633-column trace): lane = column.  A workgroup stages a
// [kOpenTileCols columns] x [kOpenTileK coefficients] tile through LDS: the global read is
// coalesced along k (one 128-byte line per column and tile), the LDS read-back is transposed, so
// every lane walks the coefficients of its own column while the powers of zeta are uniform
// scalar operands.  No cross-lane reduction is needed at all; the next tile's global loads are
// issued before the current tile is consumed.  Row pitch kOpenTileK + 1 words keeps both the
// column-wise writes and the row-wise reads free of bank conflicts.
constexpr int kOpenTileCols = kThreads, kOpenTileK = 32, kOpenPitch = kOpenTileK + 1;

__global__ __launch_bounds__(kThreads) void open_wide_kernel(const uint32_t* __restrict__ coefs, size_t coefs_stride,
                                                            int ncols, int logh, const uint32_t* __restrict__ zpow,
                                                            size_t zpow_stride, int npoints,
                                                            uint32_t* __restrict__ opened, size_t opened_stride,
                                                            size_t pt_stride) {
  __shared__ uint32_t tile[kOpenTileCols * kOpenPitch];
  const int h = 1 << logh;
  const int col0 = blockIdx.x * kOpenTileCols, b = blockIdx.y, tid = threadIdx.x;
  const uint32_t* cf = coefs + (size_t)b * coefs_stride + (size_t)col0 * h;
  const int4* z0 = reinterpret_cast<const int4*>(zpow + (size_t)b * zpow_stride);
  const int4* z1 = z0 + h;
  // loader role: 8 threads cover one column's 32 coefficients (16 bytes each), 32 columns per pass
  const int lq = tid & 7, lr = tid >> 3;
  constexpr int kPasses = kOpenTileCols / (kThreads / 8);
  uint4 stage[kPasses];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
      const int c = ps * (kThreads / 8) + lr;
      stage[ps] = col0 + c < ncols ? *reinterpret_cast<const uint4*>(cf + (size_t)c * h + k0 + lq * 4)
                                   : make_uint4(0, 0, 0, 0);
    }
  };
  int64_t acc[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[q][j] = 0;
  fetch(0);
  for (int k0 = 0; k0 < h; k0 += kOpenTileK) {
    __syncthreads();  // the previous tile has been consumed
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
      uint32_t* d = tile + (ps * (kThreads / 8) + lr) * kOpenPitch + lq * 4;
      d[0] = stage[ps].x; d[1] = stage[ps].y; d[2] = stage[ps].z; d[3] = stage[ps].w;
    }
    __syncthreads();
    if (k0 + kOpenTileK < h) fetch(k0 + kOpenTileK);
    const uint32_t* mine = tile + tid * kOpenPitch;
#pragma unroll
    for (int kk = 0; kk < kOpenTileK; ++kk) {
      const int32_t cv = fps_centre(mine[kk]);
      const int4 p0 = z0[k0 + kk];  // uniform: scalar loads
      acc[0][0] += (int64_t)cv * (int64_t)p0.x;
      acc[0][1] += (int64_t)cv * (int64_t)p0.y;
      acc[0][2] += (int64_t)cv * (int64_t)p0.z;
      acc[0][3] += (int64_t)cv * (int64_t)p0.w;
      if (npoints > 1) {
        const int4 p1 = z1[k0 + kk];
        acc[1][0] += (int64_t)cv * (int64_t)p1.x;
        acc[1][1] += (int64_t)cv * (int64_t)p1.y;
        acc[1][2] += (int64_t)cv * (int64_t)p1.z;
        acc[1][3] += (int64_t)cv * (int64_t)p1.w;
      }
      if ((kk & 7) == 7) {  // eight terms of at most p^2 / 4 since the last shrink
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[q][j] = lazy_shrink(acc[q][j]);
      }
    }
  }
  if (col0 + tid < ncols) {
    for (int q = 0; q < npoints; ++q) {
      Fp4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v.c[j] = Fp::raw(fps_canon(fps_fold(acc[q][j])));
      store_fp4(opened + (size_t)b * opened_stride + ((size_t)q * pt_stride + (size_t)(col0 + tid)) * 4, v);
    }
  }
}

// Tall matrices (the machine proof's 2^15..2^21-row chips): the transposed kernel above, with the
// coefficient range split over blockIdx.y so that a few hundred columns still make a thousand
// workgroups.  Every workgroup covers 256 columns x `klen` coefficients and leaves one partial
// extension-field sum per column and point; open_combine_kernel adds the partials (exact field
// additions, so the split does not change the result).
__device__ __forceinline__ void open_tall_body(const uint32_t* __restrict__ coefs, size_t coefs_stride, int ncols, int logh,
                                               const uint32_t* __restrict__ zpow, size_t zpow_stride, int npoints, int klen,
                                               uint32_t* __restrict__ partial, int nsplit, int bx, int split, int b) {
  __shared__ uint32_t tile[kOpenTileCols * kOpenPitch];
  // the powers of zeta of the current 32 coefficients: a 16 MB table per proof no longer lives in the
  // scalar cache, so a tile's 64 values are fetched by 64 lanes (coalesced) and broadcast from LDS
  __shared__ int4 ztile[2][kOpenTileK];
  const int h = 1 << logh;
  const int col0 = bx * kOpenTileCols, tid = threadIdx.x;
  const int kbeg = split * klen, kend = kbeg + klen;
  const uint32_t* cf = coefs + (size_t)b * coefs_stride + (size_t)col0 * h;
  const int4* z0 = reinterpret_cast<const int4*>(zpow + (size_t)b * zpow_stride);
  const int4* z1 = z0 + h;
  const int lq = tid & 7, lr = tid >> 3;
  constexpr int kPasses = kOpenTileCols / (kThreads / 8);
  uint4 stage[kPasses];
  int4 zstage = make_int4(0, 0, 0, 0);
  const int zq = tid >> 5, zk = tid & 31;  // threads 0..63: point zq, coefficient zk of the tile
  auto fetch = [&](int k0) {
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
      const int c = ps * (kThreads / 8) + lr;
      stage[ps] = col0 + c < ncols ? *reinterpret_cast<const uint4*>(cf + (size_t)c * h + k0 + lq * 4)
                                   : make_uint4(0, 0, 0, 0);
    }
    if (tid < 2 * kOpenTileK && zq < npoints) zstage = (zq ? z1 : z0)[k0 + zk];
  };
  int64_t acc[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[q][j] = 0;
  fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += kOpenTileK) {
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
      uint32_t* d = tile + (ps * (kThreads / 8) + lr) * kOpenPitch + lq * 4;
      d[0] = stage[ps].x; d[1] = stage[ps].y; d[2] = stage[ps].z; d[3] = stage[ps].w;
    }
    if (tid < 2 * kOpenTileK) ztile[zq][zk] = zstage;
    __syncthreads();
    if (k0 + kOpenTileK < kend) fetch(k0 + kOpenTileK);
    const uint32_t* mine = tile + tid * kOpenPitch;
#pragma unroll 8
    for (int kk = 0; kk < kOpenTileK; ++kk) {
      const int32_t cv = fps_centre(mine[kk]);
      const int4 p0 = ztile[0][kk];
      acc[0][0] += (int64_t)cv * (int64_t)p0.x;
      acc[0][1] += (int64_t)cv * (int64_t)p0.y;
      acc[0][2] += (int64_t)cv * (int64_t)p0.z;
      acc[0][3] += (int64_t)cv * (int64_t)p0.w;
      if (npoints > 1) {
        const int4 p1 = ztile[1][kk];
        acc[1][0] += (int64_t)cv * (int64_t)p1.x;
        acc[1][1] += (int64_t)cv * (int64_t)p1.y;
        acc[1][2] += (int64_t)cv * (int64_t)p1.z;
        acc[1][3] += (int64_t)cv * (int64_t)p1.w;
      }
      if ((kk & 7) == 7) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[q][j] = lazy_shrink(acc[q][j]);
      }
    }
  }
  if (col0 + tid < ncols) {
    for (int q = 0; q < npoints; ++q) {
      Fp4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v.c[j] = Fp::raw(fps_canon(fps_fold(acc[q][j])));
      store_fp4(partial + ((((size_t)b * 2 + q) * nsplit + split) * ncols + col0 + tid) * 4, v);
    }
  }
}
__global__ __launch_bounds__(kThreads) void open_tall_kernel(const uint32_t* __restrict__ coefs, size_t coefs_stride,
                                                            int ncols, int logh, const uint32_t* __restrict__ zpow,
                                                            size_t zpow_stride, int npoints, int klen,
                                                            uint32_t* __restrict__ partial, int nsplit) {
  open_tall_body(coefs, coefs_stride, ncols, logh, zpow, zpow_stride, npoints, klen, partial, nsplit, blockIdx.x, blockIdx.y, blockIdx.z);
}
// Sixteen lanes per (column, point): each adds every sixteenth partial, then the lanes are added (exact field additions, in
// any order).  One lane per column walking up to 128 partials one dependent load after the other took 60-70 us of a single
// proof's opening stage, fifty times over.
__device__ __forceinline__ void open_combine_body(const uint32_t* __restrict__ partial, int ncols, int npoints, int nsplit,
                                                  uint32_t* __restrict__ opened, size_t opened_stride, size_t pt_stride, int bx, int b) {
  const int l = threadIdx.x & 15, item = bx * (kThreads / 16) + (threadIdx.x >> 4);
  const bool ok = item < ncols * npoints;
  const int q = ok ? item / ncols : 0, col = ok ? item % ncols : 0;
  Fp4 s = Fp4::zero();
  if (ok)
    for (int sp = l; sp < nsplit; sp += 16) s += load_fp4(partial + ((((size_t)b * 2 + q) * nsplit + sp) * ncols + col) * 4);
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    Fp4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.c[j] = Fp::raw((uint32_t)__shfl_xor((int)s.c[j].v, off, 64));
    s += o;
  }
  if (ok && l == 0) store_fp4(opened + (size_t)b * opened_stride + ((size_t)q * pt_stride + (size_t)col) * 4, s);
}
__global__ __launch_bounds__(kThreads) void open_combine_kernel(const uint32_t* __restrict__ partial, int ncols, int npoints,
                                                               int nsplit, uint32_t* __restrict__ opened,
                                                               size_t opened_stride, size_t pt_stride) {
  open_combine_body(partial, ncols, npoints, nsplit, opened, opened_stride, pt_stride, blockIdx.x, blockIdx.y);
}
// Narrow tall matrices, tiled: a workgroup covers CW columns (the whole matrix when it has at most 64) x `klen`
// evaluations, so the two weight tables are read once per matrix instead of once per four columns (a lane-per-evaluation
// kernel reads 32 bytes of weights for every 16 bytes of columns; at batch 128 the tables are 2 GB and live in HBM).  A tile is
// [CW columns][TK = 2048 / CW evaluations] in LDS; lane (c = tid % CW, phase = tid / CW) walks column c's eight
// evaluations phase, phase + 256 / CW, ... of the tile with the weights broadcast from LDS; the phases of a column are
// added at the end.  Partial sums in open_tall_kernel's layout.
template <int CW>
__device__ __forceinline__ void open_narrow_body(const uint32_t* __restrict__ coefs, size_t coefs_stride, int ncols, int logh,
                                                 const uint32_t* __restrict__ zpow, size_t zpow_stride, int npoints, int klen,
                                                 uint32_t* __restrict__ partial, int nsplit, int bx, int split, int b) {
  constexpr int TK = 2048 / CW, kPitch = TK + 1, kPhases = kThreads / CW, kQuads = TK / 4;
  static_assert(kThreads == 256 && CW >= 8 && CW <= 64, "tile geometry");
  __shared__ uint32_t tile[CW * kPitch];
  __shared__ int4 ztile[2][TK];
  __shared__ uint32_t red[2][kThreads][4];
  const int h = 1 << logh;
  const int col0 = bx * CW, tid = threadIdx.x;
  const int kbeg = split * klen, kend = kbeg + klen;
  const uint32_t* cf = coefs + (size_t)b * coefs_stride + (size_t)col0 * h;
  const int4* z0 = reinterpret_cast<const int4*>(zpow + (size_t)b * zpow_stride);
  const int4* z1 = z0 + h;
  uint4 stage[2];
  int4 zstage[2] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};
  auto fetch = [&](int k0) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int e = ps * kThreads + tid, c = e / kQuads, q4 = e % kQuads;
      stage[ps] = col0 + c < ncols ? *reinterpret_cast<const uint4*>(cf + (size_t)c * h + k0 + q4 * 4) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int e = ps * kThreads + tid;  // point e / TK, evaluation e % TK
      if (e < 2 * TK && e / TK < npoints) zstage[ps] = (e / TK ? z1 : z0)[k0 + e % TK];
    }
  };
  int64_t acc[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[q][j] = 0;
  const int c = tid % CW, ph = tid / CW;
  fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += TK) {
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int e = ps * kThreads + tid;
      uint32_t* d = tile + (e / kQuads) * kPitch + (e % kQuads) * 4;
      d[0] = stage[ps].x; d[1] = stage[ps].y; d[2] = stage[ps].z; d[3] = stage[ps].w;
      if (e < 2 * TK) ztile[e / TK][e % TK] = zstage[ps];
    }
    __syncthreads();
    if (k0 + TK < kend) fetch(k0 + TK);
    const uint32_t* mine = tile + c * kPitch;
#pragma unroll
    for (int i = 0; i < TK / kPhases; ++i) {
      const int kk = ph + i * kPhases;
      const int32_t cv = fps_centre(mine[kk]);
      const int4 p0 = ztile[0][kk];
      acc[0][0] += (int64_t)cv * (int64_t)p0.x;
      acc[0][1] += (int64_t)cv * (int64_t)p0.y;
      acc[0][2] += (int64_t)cv * (int64_t)p0.z;
      acc[0][3] += (int64_t)cv * (int64_t)p0.w;
      if (npoints > 1) {
        const int4 p1 = ztile[1][kk];
        acc[1][0] += (int64_t)cv * (int64_t)p1.x;
        acc[1][1] += (int64_t)cv * (int64_t)p1.y;
        acc[1][2] += (int64_t)cv * (int64_t)p1.z;
        acc[1][3] += (int64_t)cv * (int64_t)p1.w;
      }
    }
    static_assert(TK / kPhases == 8, "eight terms between shrinks");
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[q][j] = lazy_shrink(acc[q][j]);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[q][tid][j] = fps_canon(fps_fold(acc[q][j]));
  __syncthreads();
  if (ph == 0 && col0 + c < ncols) {
    for (int q = 0; q < npoints; ++q) {
      Fp4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v.c[j] = Fp::raw(red[q][c][j]);
      for (int o = 1; o < kPhases; ++o) {
        Fp4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w.c[j] = Fp::raw(red[q][o * CW + c][j]);
        v += w;
      }
      store_fp4(partial + ((((size_t)b * 2 + q) * nsplit + split) * ncols + col0 + c) * 4, v);
    }
  }
}

template <int CW>
__global__ __launch_bounds__(kThreads) void open_narrow_kernel(const uint32_t* __restrict__ coefs, size_t coefs_stride,
                                                              int ncols, int logh, const uint32_t* __restrict__ zpow,
                                                              size_t zpow_stride, int npoints, int klen,
                                                              uint32_t* __restrict__ partial, int nsplit) {
  open_narrow_body<CW>(coefs, coefs_stride, ncols, logh, zpow, zpow_stride, npoints, klen, partial, nsplit, blockIdx.x, blockIdx.y,
                       blockIdx.z);
}

// ---- a single proof's opening stage: ONE launch per kind of kernel over a table of tasks (OpenTask, kernels.h) instead of
// two launches per matrix - 140 launches of 5-20 us of work each were 1.2 ms of a 10 ms proof.  A workgroup finds its task
// from its index (the tasks' first workgroups are a prefix sum, in launch order); what it computes is the single-task
// kernel's body, so the sums are the same words. ----
__device__ __forceinline__ const OpenTask& open_task_of(const OpenTask* __restrict__ tasks, int n, int blk, int OpenTask::*first) {
  int t = 0;
  while (t + 1 < n && blk >= tasks[t + 1].*first) ++t;
  return tasks[t];
}
__global__ __launch_bounds__(kThreads) void open_multi_short_kernel(const OpenTask* __restrict__ tasks, int n, size_t opened_stride) {
  const OpenTask& t = open_task_of(tasks, n, blockIdx.x, &OpenTask::blk0);
  const int local = blockIdx.x - t.blk0, groups = (t.ncols + kOpenCols - 1) / kOpenCols;
  open_body(t.evals, t.cstride, t.ncols, t.logh, t.table, t.zstride, t.npts, t.dst, opened_stride, t.pts, local % groups, local / groups);
}
template <int CW>
__global__ __launch_bounds__(kThreads) void open_multi_narrow_kernel(const OpenTask* __restrict__ tasks, int n) {
  const OpenTask& t = open_task_of(tasks, n, blockIdx.x, &OpenTask::blk0);
  const int local = blockIdx.x - t.blk0;
  open_narrow_body<CW>(t.evals, t.cstride, t.ncols, t.logh, t.table, t.zstride, t.npts, t.klen, t.partial, t.nsplit, 0,
                       local % t.nsplit, local / t.nsplit);
}
__global__ __launch_bounds__(kThreads) void open_multi_tall_kernel(const OpenTask* __restrict__ tasks, int n) {
  const OpenTask& t = open_task_of(tasks, n, blockIdx.x, &OpenTask::blk0);
  const int local = blockIdx.x - t.blk0, tiles = (t.ncols + kOpenTileCols - 1) / kOpenTileCols;
  open_tall_body(t.evals, t.cstride, t.ncols, t.logh, t.table, t.zstride, t.npts, t.klen, t.partial, t.nsplit, local % tiles,
                 (local / tiles) % t.nsplit, local / (tiles * t.nsplit));
}
__global__ __launch_bounds__(kThreads) void open_multi_combine_kernel(const OpenTask* __restrict__ tasks, int n, size_t opened_stride) {
  const OpenTask& t = open_task_of(tasks, n, blockIdx.x, &OpenTask::cblk0);
  const int local = blockIdx.x - t.cblk0, groups = (t.ncols * t.npts + kThreads / 16 - 1) / (kThreads / 16);
  open_combine_body(t.partial, t.ncols, t.npts, t.nsplit, t.dst, opened_stride, t.pts, local % groups, local / groups);
}

// A batch of fewer than eight proofs splits finer (a single proof would otherwise run four workgroups of 4 096 evaluations
// each): up to eight times as many splits, so that batch x splits never exceeds what eight proofs take - which is what the
// scratch is sized for (open_tall_scratch_words) - and no split shorter than 256 evaluations (the narrow kernel's tile).
static int open_nsplit(int ncols, int logh, int batch) {
  const int h = 1 << logh;
  const int base = ncols >= 64 ? std::max(1, std::min(128, h / 4096)) : std::max(1, std::min(128, h / 2048));
  if (batch >= 8) return base;
  return std::max(base, std::min({128, h / 256, base * (8 / std::max(batch, 1))}));
}
size_t open_tall_scratch_words(int ncols, int logh, int batch) {
  return (size_t)std::max(batch, 8) * 2 * open_nsplit(ncols, logh, 8) * ncols * 4;
}
void launch_open_tall(hipStream_t stream, const uint32_t* coefs_br, size_t coefs_stride, int ncols, int logh,
                      const uint32_t* zpow_br, size_t zpow_stride, int npoints, uint32_t* opened, size_t opened_stride,
                      size_t pt_stride, uint32_t* scratch, int batch) {
  const int h = 1 << logh;
  const int nsplit = open_nsplit(ncols, logh, batch), klen = h / nsplit;
  if (ncols >= 64)
    hipLaunchKernelGGL(open_tall_kernel, dim3((ncols + kOpenTileCols - 1) / kOpenTileCols, nsplit, batch), dim3(kThreads), 0,
                       stream, coefs_br, coefs_stride, ncols, logh, zpow_br, zpow_stride, npoints, klen, scratch, nsplit);
  else {
    const dim3 grid(1, nsplit, batch), block(kThreads);
#define ZKSP_OPEN_NARROW(CW) \
  hipLaunchKernelGGL(open_narrow_kernel<CW>, grid, block, 0, stream, coefs_br, coefs_stride, ncols, logh, zpow_br, zpow_stride, \
                     npoints, klen, scratch, nsplit)
    if (ncols > 32) ZKSP_OPEN_NARROW(64);
    else if (ncols > 16) ZKSP_OPEN_NARROW(32);
    else if (ncols > 8) ZKSP_OPEN_NARROW(16);
    else ZKSP_OPEN_NARROW(8);
#undef ZKSP_OPEN_NARROW
  }
  hipLaunchKernelGGL(open_combine_kernel, dim3((ncols * npoints + kThreads / 16 - 1) / (kThreads / 16), batch), dim3(kThreads), 0,
                     stream, scratch, ncols, npoints, nsplit, opened, opened_stride, pt_stride);
}

// Fills in a task's kind, split, partial-sum need and workgroup counts exactly as launch_open / launch_open_tall choose them.
// Returns the words of partial sums the task needs (0 for the short kind).
size_t open_task_plan(OpenTask* t, int batch) {
  const int h = 1 << t->logh;
  if (t->logh < 12) {
    t->kind = 0;
    t->nsplit = 1;
    t->klen = h;
    t->blocks = ((t->ncols + kOpenCols - 1) / kOpenCols) * batch;
    t->cblocks = 0;
    return 0;
  }
  t->nsplit = open_nsplit(t->ncols, t->logh, batch);
  t->klen = h / t->nsplit;
  if (t->ncols >= 64) {
    t->kind = 5;
    t->blocks = ((t->ncols + kOpenTileCols - 1) / kOpenTileCols) * t->nsplit * batch;
  } else {
    t->kind = t->ncols > 32 ? 4 : t->ncols > 16 ? 3 : t->ncols > 8 ? 2 : 1;
    t->blocks = t->nsplit * batch;
  }
  t->cblocks = ((t->ncols * t->npts + kThreads / 16 - 1) / (kThreads / 16)) * batch;
  return (size_t)batch * 2 * t->nsplit * t->ncols * 4;
}
// tasks: device array sorted by kind (0, 1..4, 5), `first[k]` / `count[k]` the tasks of kind k, blk0 counted per kind,
// cblk0 over the kinds 1..5 together
void launch_open_multi(hipStream_t stream, const OpenTask* tasks, const int first[6], const int count[6], const int blocks[6],
                       int combine_blocks, size_t opened_stride) {
  if (count[0])
    hipLaunchKernelGGL(open_multi_short_kernel, dim3(blocks[0]), dim3(kThreads), 0, stream, tasks + first[0], count[0], opened_stride);
  if (count[1]) hipLaunchKernelGGL(open_multi_narrow_kernel<8>, dim3(blocks[1]), dim3(kThreads), 0, stream, tasks + first[1], count[1]);
  if (count[2]) hipLaunchKernelGGL(open_multi_narrow_kernel<16>, dim3(blocks[2]), dim3(kThreads), 0, stream, tasks + first[2], count[2]);
  if (count[3]) hipLaunchKernelGGL(open_multi_narrow_kernel<32>, dim3(blocks[3]), dim3(kThreads), 0, stream, tasks + first[3], count[3]);
  if (count[4]) hipLaunchKernelGGL(open_multi_narrow_kernel<64>, dim3(blocks[4]), dim3(kThreads), 0, stream, tasks + first[4], count[4]);
  if (count[5]) hipLaunchKernelGGL(open_multi_tall_kernel, dim3(blocks[5]), dim3(kThreads), 0, stream, tasks + first[5], count[5]);
  const int nc = count[1] + count[2] + count[3] + count[4] + count[5];
  if (nc)
    hipLaunchKernelGGL(open_multi_combine_kernel, dim3(combine_blocks), dim3(kThreads), 0, stream, tasks + first[1], nc, opened_stride);
}

void launch_open(hipStream_t stream, const uint32_t* coefs_br, size_t coefs_stride, int ncols, int logh,
                 const uint32_t* zpow_br, size_t zpow_stride, int npoints, uint32_t* opened, size_t opened_stride,
                 size_t pt_stride, int batch) {
  // The transposed kernel runs few, long workgroups (one per 256 columns, every tile in
  // sequence): it wins once those fill the chip several times over (large batches) and loses
  // badly on a single proof, where the lane-per-k kernel offers 60x more workgroups.
  const long wide_blocks = (long)((ncols + kOpenTileCols - 1) / kOpenTileCols) * batch;
  if (ncols >= 64 && logh >= 5 && wide_blocks >= 4 * 256) {
    hipLaunchKernelGGL(open_wide_kernel, dim3((ncols + kOpenTileCols - 1) / kOpenTileCols, batch), dim3(kThreads), 0,
                       stream, coefs_br, coefs_stride, ncols, logh, zpow_br, zpow_stride, npoints, opened, opened_stride,
                       pt_stride);
    return;
  }
  hipLaunchKernelGGL(open_kernel, dim3((ncols + kOpenCols - 1) / kOpenCols, batch), dim3(kThreads), 0, stream, coefs_br,
                     coefs_stride, ncols, logh, zpow_br, zpow_stride, npoints, opened, opened_stride, pt_stride);
}

#ifdef ZKSP_COMPONENT  // the round-1 keccak-chip component path only (include/zksp_component.h)
// ===========================================================================
// reduced openings (DEEP quotient) over the LDE domain
// ===========================================================================
constexpr int kReduceChunk = 128;

__global__ __launch_bounds__(kThreads) void reduce_bsum_kernel(ReduceArgs a) {
  __shared__ Fp4 red[kThreads / 64];
  const int b = blockIdx.x, W = a.width;
  const uint32_t* ap = a.af_pows + (size_t)b * (2 * W + 16) * 4;
  const uint32_t* op = a.opened + (size_t)b * a.opened_stride;
  Fp4 s0 = Fp4::zero(), s1 = Fp4::zero(), s2 = Fp4::zero(), s3 = Fp4::zero(), s4 = Fp4::zero();
  for (int i = threadIdx.x; i < W; i += kThreads) {
    Fp4 p = load_fp4(ap + (size_t)i * 4);
    s0 += p * load_fp4(op + (size_t)i * 4);
    s1 += p * load_fp4(op + (size_t)(W + i) * 4);
  }
  if (threadIdx.x < 8) s2 = load_fp4(ap + (size_t)threadIdx.x * 4) * load_fp4(op + (size_t)(2 * W + threadIdx.x) * 4);
  if (threadIdx.x < 4) {
    s3 = load_fp4(ap + (size_t)threadIdx.x * 4) * load_fp4(op + (size_t)(2 * W + 8 + threadIdx.x) * 4);
    s4 = load_fp4(ap + (size_t)threadIdx.x * 4) * load_fp4(op + (size_t)(2 * W + 12 + threadIdx.x) * 4);
  }
  Fp4 r0 = block_sum(s0, red), r1 = block_sum(s1, red), r2 = block_sum(s2, red), r3 = block_sum(s3, red),
      r4 = block_sum(s4, red);
  if (threadIdx.x == 0) {
    store_fp4(a.bsum + ((size_t)b * 5 + 0) * 4, r0);
    store_fp4(a.bsum + ((size_t)b * 5 + 1) * 4, r1);
    store_fp4(a.bsum + ((size_t)b * 5 + 2) * 4, r2);
    store_fp4(a.bsum + ((size_t)b * 5 + 3) * 4, r3);
    store_fp4(a.bsum + ((size_t)b * 5 + 4) * 4, r4);
  }
}

// Lane = four consecutive LDE points (one 16-byte load per column, so a wave keeps 1 KB per
// load in flight); blockIdx.y = a chunk of kReduceChunk columns.  Lazy dot product: the powers
// of alpha are uniform, so centring them is scalar work; the LDE words stay canonical,
// |term| < p^2 / 2, four terms between shrinks.
__global__ __launch_bounds__(kThreads) void reduce_partial_kernel(ReduceArgs a, int nchunks) {
  const int h = 1 << a.logh, n = 2 * h;
  const int pt = (blockIdx.x * kThreads + threadIdx.x) * 4;
  if (pt >= n) return;
  const int chunk = blockIdx.y, b = blockIdx.z, W = a.width;
  const int i0 = chunk * kReduceChunk, i1 = min(W, i0 + kReduceChunk);
  const uint32_t* col = a.lde_t + (size_t)b * W * n + pt;
  const uint32_t* ap = a.af_pows + (size_t)b * (2 * W + 16) * 4;
  int64_t acc[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[q][j] = 0;
  for (int i = i0; i < i1; i += 4) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = i + u < i1 ? *reinterpret_cast<const uint4*>(col + (size_t)(i + u) * n) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t* al = ap + (size_t)min(i + u, i1 - 1) * 4;
      const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int32_t alpha = fps_centre(al[j]);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q][j] += (int64_t)alpha * (int64_t)w[q];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[q][j] = lazy_shrink(acc[q][j]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    Fp4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r.c[j] = Fp::raw(fps_canon(fps_fold(acc[q][j])));
    store_fp4(a.partial + (((size_t)b * nchunks + chunk) * n + pt + q) * 4, r);
  }
}

__global__ __launch_bounds__(kThreads) void reduce_final_kernel(ReduceArgs a, int nchunks) {
  const int h = 1 << a.logh, n = 2 * h;
  const int pt = blockIdx.x * kThreads + threadIdx.x;
  if (pt >= n) return;
  const int b = blockIdx.y, W = a.width;
  const uint32_t* ap = a.af_pows + (size_t)b * (2 * W + 16) * 4;
  Fp4 st = Fp4::zero();
  for (int ch = 0; ch < nchunks; ++ch) st += load_fp4(a.partial + (((size_t)b * nchunks + ch) * n + pt) * 4);
  Fp4 sq = Fp4::zero();
  const uint32_t* q = a.lde_q + (size_t)b * 8 * n + pt;
#pragma unroll
  for (int i = 0; i < 8; ++i) sq += load_fp4(ap + (size_t)i * 4) * Fp::raw(q[(size_t)i * n]);
  Fp4 sp = Fp4::zero();
  const uint32_t* pp = a.lde_p + (size_t)b * 4 * n + pt;
#pragma unroll
  for (int i = 0; i < 4; ++i) sp += load_fp4(ap + (size_t)i * 4) * Fp::raw(pp[(size_t)i * n]);
  const Fp4 b0 = load_fp4(a.bsum + ((size_t)b * 5 + 0) * 4), b1 = load_fp4(a.bsum + ((size_t)b * 5 + 1) * 4),
            b2 = load_fp4(a.bsum + ((size_t)b * 5 + 2) * 4), b3 = load_fp4(a.bsum + ((size_t)b * 5 + 3) * 4),
            b4 = load_fp4(a.bsum + ((size_t)b * 5 + 4) * 4);
  const Fp4 zeta = load_fp4(a.zeta + (size_t)b * 4);
  const Fp4 zeta_next = zeta * Fp::raw(a.w_h);
  const Fp4 x = Fp4::from_base(Fp::raw(a.xs[pt]));
  const Fp4 d0 = (x - zeta).inv(), d1 = (x - zeta_next).inv();
  Fp4 g = (st - b0) * d0;
  g += load_fp4(ap + (size_t)W * 4) * (st - b1) * d1;
  g += load_fp4(ap + (size_t)(2 * W) * 4) * (sq - b2) * d0;
  g += load_fp4(ap + (size_t)(2 * W + 8) * 4) * (sp - b3) * d0;
  g += load_fp4(ap + (size_t)(2 * W + 12) * 4) * (sp - b4) * d1;
  store_fp4(a.out + (size_t)b * a.out_stride + (size_t)pt * 4, g);
}

void launch_reduce_openings(hipStream_t stream, const ReduceArgs& a) {
  const int n = 2 << a.logh;
  const int blocks = (n + kThreads - 1) / kThreads;
  const int nchunks = (a.width + kReduceChunk - 1) / kReduceChunk;
  hipLaunchKernelGGL(reduce_bsum_kernel, dim3(a.batch), dim3(kThreads), 0, stream, a);
  hipLaunchKernelGGL(reduce_partial_kernel, dim3((n / 4 + kThreads - 1) / kThreads, nchunks, a.batch), dim3(kThreads), 0,
                     stream, a, nchunks);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(blocks, a.batch), dim3(kThreads), 0, stream, a, nchunks);
}
int reduce_nchunks(int width) { return (width + kReduceChunk - 1) / kReduceChunk; }

#endif  // ZKSP_COMPONENT

// ===========================================================================
// FRI fold
// ===========================================================================
__global__ __launch_bounds__(kThreads) void fri_fold_kernel(const uint32_t* __restrict__ in, size_t in_stride,
                                                           uint32_t* __restrict__ out, size_t out_stride,
                                                           const uint32_t* __restrict__ beta, size_t beta_stride,
                                                           const uint32_t* __restrict__ tw_inv, int tw_shift,
                                                           uint32_t xinv0, uint32_t xinv1, int loghk,
                                                           const uint32_t* __restrict__ join, size_t join_stride) {
  const int hk = 1 << loghk, half = hk >> 1;
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= hk) return;  // 2 cosets * half
  const int b = blockIdx.y;
  const int c = i >= half ? 1 : 0, m = i - c * half;
  const uint32_t* f = in + (size_t)b * in_stride;
  Fp4 lo = load_fp4(f + ((size_t)c * hk + m) * 4), hi = load_fp4(f + ((size_t)c * hk + m + half) * 4);
  const Fp inv2 = Fp::raw(cmonty((kP + 1) / 2));
  Fp xinv = Fp::raw(c ? xinv1 : xinv0) * Fp::raw(tw_inv[(size_t)m << tw_shift]);
  Fp4 be = load_fp4(beta + (size_t)b * beta_stride);
  Fp4 r = (lo + hi) * inv2 + be * ((lo - hi) * (inv2 * xinv));
  // (`join`: the reduced opening of the height the folded layer has reached, same layout: added here instead of by a launch)
  if (join) r += load_fp4(join + (size_t)b * join_stride + ((size_t)c * half + m) * 4);
  store_fp4(out + (size_t)b * out_stride + ((size_t)c * half + m) * 4, r);
}

void launch_fri_fold(hipStream_t stream, const uint32_t* in, size_t in_stride, uint32_t* out, size_t out_stride,
                     const uint32_t* beta, size_t beta_stride, const uint32_t* tw_inv, int tw_shift, uint32_t xinv0,
                     uint32_t xinv1, int loghk, int batch, const uint32_t* join, size_t join_stride) {
  const int hk = 1 << loghk;
  hipLaunchKernelGGL(fri_fold_kernel, dim3((hk + kThreads - 1) / kThreads, batch), dim3(kThreads), 0, stream, in,
                     in_stride, out, out_stride, beta, beta_stride, tw_inv, tw_shift, xinv0, xinv1, loghk, join, join_stride);
}

// ===========================================================================
// duplex challenger: one lane per proof
// ===========================================================================
// The sponge of one proof lives on 16 adjacent lanes (element e of the state, of
// the input buffer and of the output buffer in lane e), so that every permutation
// of this strictly sequential part runs in its cooperative, short-latency form.
struct Ch {
  Fp st, in, out;
  int n_in, n_out;  // identical on the 16 lanes of a row
  int e;
  CoopConsts cc;
};
__device__ __forceinline__ void ch_load(const DevChallenger* d, Ch& c, int e, const P2Consts* k) {
  c.e = e;
  c.cc = coop_load_consts(k, e);
  c.st = Fp::raw(d->state[e]);
  c.in = Fp::raw(d->inbuf[e & 7]);
  c.out = Fp::raw(d->outbuf[e & 7]);
  c.n_in = d->n_in;
  c.n_out = d->n_out;
}
__device__ __forceinline__ void ch_store(DevChallenger* d, const Ch& c) {
  d->state[c.e] = c.st.v;
  if (c.e < 8) {
    d->inbuf[c.e] = c.in.v;
    d->outbuf[c.e] = c.out.v;
  }
  if (c.e == 0) {
    d->n_in = c.n_in;
    d->n_out = c.n_out;
  }
}
__device__ __forceinline__ void ch_duplex(Ch& c, const P2Consts* k) {
  if (c.e < c.n_in) c.st = c.in;
  c.n_in = 0;
  c.st = p2_permute_coop(c.st, c.cc, k);
  c.out = c.st;
  c.n_out = 8;
}
__device__ __forceinline__ void ch_observe(Ch& c, Fp x, const P2Consts* k) {
  c.n_out = 0;
  if (c.e == c.n_in) c.in = x;
  c.n_in++;
  if (c.n_in == 8) ch_duplex(c, k);
}
__device__ __forceinline__ Fp ch_sample(Ch& c, const P2Consts* k) {
  if (c.n_in != 0 || c.n_out == 0) ch_duplex(c, k);
  c.n_out--;
  return Fp::raw((uint32_t)__shfl((int)c.out.v, c.n_out, 16));
}

constexpr int kChThreads = 64;  // four proofs per wave

__global__ __launch_bounds__(kChThreads) void ch_init_kernel(DevChallenger* ch, const uint32_t* __restrict__ init_obs,
                                                            int n_obs, int batch, const P2Consts* __restrict__ k) {
  const int t = blockIdx.x * kChThreads + threadIdx.x, b = t >> 4, e = t & 15;
  if (b >= batch) return;
  Ch c;
  c.e = e;
  c.cc = coop_load_consts(k, e);
  c.st = c.in = c.out = Fp::zero();
  c.n_in = 0;
  c.n_out = 0;
  for (int i = 0; i < n_obs; ++i) ch_observe(c, Fp::from_canonical(init_obs[(size_t)b * n_obs + i]), k);
  ch_store(ch + b, c);
}

__global__ __launch_bounds__(kChThreads) void ch_observe_sample_kernel(DevChallenger* ch,
                                                                      const uint32_t* __restrict__ obs,
                                                                      size_t obs_stride, int n_obs,
                                                                      uint32_t* __restrict__ out, size_t out_stride,
                                                                      int n_ext, int batch,
                                                                      const P2Consts* __restrict__ k, int align) {
  const int t = blockIdx.x * kChThreads + threadIdx.x, b = t >> 4, e = t & 15;
  if (b >= batch) return;
  Ch c;
  ch_load(ch + b, c, e, k);
  for (int i = 0; i < n_obs; ++i) ch_observe(c, Fp::raw(obs[(size_t)b * obs_stride + i]), k);
  // machine proofs (format v16): a phase of the transcript ends on a block boundary - a pending block is zero-filled
  if (align)
    while (c.n_in != 0) ch_observe(c, Fp::zero(), k);
  for (int i = 0; i < 4 * n_ext; ++i) {
    Fp v = ch_sample(c, k);
    if (e == 0) out[(size_t)b * out_stride + i] = v.v;
  }
  ch_store(ch + b, c);
}

// ===========================================================================
// FRI commit phase, layers of at most 512 leaves: commit, transcript and fold of EVERY such
// layer in one launch, one workgroup per proof.  Separately these are three launches per layer
// (27 for a 2^11 trace), each a short dependent step; a single proof spent a quarter of its time
// there.  The work is unchanged: coop_fri_commit_block per layer, the proof's sponge advanced by
// the first 16 lanes (which keep its state in registers across layers), the fold by all threads.
// ===========================================================================
__global__ __launch_bounds__(kTopThreads) void fri_tail_kernel(FriTailArgs a, const P2Consts* __restrict__ k) {
  __shared__ uint32_t beta_s[4];
  const int b = blockIdx.x, tid = threadIdx.x, e = tid & 15;
  const CoopConsts cc = coop_load_consts(k, e);
  Ch c;
  if (tid < 16) ch_load(a.ch + b, c, e, k);
  uint32_t* layers = a.layers + (size_t)b * a.layer_stride;
  uint32_t* trees = a.trees + (size_t)b * a.tree_stride;
  size_t loff = a.loff_start, toff = a.toff_start;
  for (int kk = a.k_start; kk < a.logh; ++kk) {
    const int loghk = a.logh - kk, hk = 1 << loghk, half = hk >> 1;
    uint32_t* f = layers + loff;
    uint32_t* t = trees + toff * 8;
    coop_fri_commit_block(f, t, loghk, cc, k);  // ends with a barrier: the root is visible
    if (tid < 16) {
      const uint32_t* root = t + (size_t)(2 * hk - 2) * 8;
      for (int i = 0; i < 8; ++i) ch_observe(c, Fp::raw(root[i]), k);
      for (int i = 0; i < 4; ++i) {
        const Fp v = ch_sample(c, k);
        if (e == 0) {
          beta_s[i] = v.v;
          a.betas[(size_t)b * a.beta_stride + (size_t)kk * 4 + i] = v.v;
        }
      }
    }
    __syncthreads();
    Fp4 be;
#pragma unroll
    for (int j = 0; j < 4; ++j) be.c[j] = Fp::raw(beta_s[j]);
    const Fp inv2 = Fp::raw(cmonty((kP + 1) / 2));
    uint32_t* out = f + (size_t)2 * hk * 4;
    for (int i = tid; i < hk; i += kTopThreads) {  // 2 cosets * half outputs (fri_fold_kernel's arithmetic)
      const int cset = i >= half ? 1 : 0, m = i - cset * half;
      const Fp4 lo = load_fp4(f + ((size_t)cset * hk + m) * 4), hi = load_fp4(f + ((size_t)cset * hk + m + half) * 4);
      const Fp xinv = Fp::raw(a.xinv[2 * kk + cset]) * Fp::raw(a.tw_inv[(size_t)m << kk]);
      Fp4 r = (lo + hi) * inv2 + be * ((lo - hi) * (inv2 * xinv));
      if (a.join[kk]) r += load_fp4(a.join[kk] + ((size_t)b * hk + (size_t)cset * half + m) * 4);
      store_fp4(out + ((size_t)cset * half + m) * 4, r);
    }
    __syncthreads();  // the next layer is complete, and beta_s may be rewritten
    loff += (size_t)2 * hk * 4;
    toff += (size_t)2 * hk - 1;
  }
  if (tid < 16) ch_store(a.ch + b, c);
}

void launch_fri_tail(hipStream_t stream, const FriTailArgs& a, int batch, const P2Consts* consts) {
  hipLaunchKernelGGL(fri_tail_kernel, dim3(batch), dim3(kTopThreads), 0, stream, a, consts);
}

// Proof-of-work search: smallest w such that, after observing w, the next sample
// has `bits` low zero bits.  observe(w) then sample() on a duplex sponge always
// amounts to: overlay the pending inputs and w on the rate part, permute once,
// read word 7.
//
// The candidates are covered by a FIXED sequence of launches, one 2^14-wide range
// per launch for every proof of the batch; a workgroup returns at once when a
// smaller witness is already known, so once a proof is done its share of the later
// launches costs a load and a compare.  Every candidate below the final minimum
// has been tested by construction, hence the result is the same smallest witness
// the sequential CPU search returns, whatever the scheduling.  Expected work is
// 2^bits + 2^13 permutations per proof.
constexpr uint32_t kGrindRange = 1u << 14;
constexpr int kGrindLaunches = 64;  // covers 2^20 candidates: miss probability e^-16 at 16 bits

// (`pad`: the words behind the witness are zero-filled - machine proofs since format v16 - instead of left as they are)
__device__ __forceinline__ bool grind_try(const DevChallenger* d, uint32_t w, uint32_t mask, const P2Consts* k, int pad) {
  const int pos = d->n_in;  // 0..7
  Fp st[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) st[i] = Fp::raw(d->state[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < pos) st[i] = Fp::raw(d->inbuf[i]);
  const Fp wm = Fp::from_canonical(w);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i == pos) st[i] = wm;
    else if (pad && i > pos) st[i] = Fp::zero();
  p2_permute(st, k);
  return (st[7].to_canonical() & mask) == 0;
}

__global__ __launch_bounds__(kThreads) void ch_grind_kernel(const DevChallenger* __restrict__ ch,
                                                           uint32_t* __restrict__ witness, int bits, uint32_t range_lo,
                                                           const P2Consts* __restrict__ k, int pad) {
  const int b = blockIdx.y;
  const uint32_t lo = range_lo + blockIdx.x * kThreads;
  if (witness[b] < lo) return;  // settled by an earlier launch (or an earlier chunk of this one)
  const uint32_t w = lo + threadIdx.x;
  if (grind_try(ch + b, w, (1u << bits) - 1, k, pad)) atomicMin(&witness[b], w);
}

// continuation for the (practically unreachable) case that 2^20 candidates all failed
__global__ __launch_bounds__(kThreads) void ch_grind_tail_kernel(const DevChallenger* __restrict__ ch,
                                                                uint32_t* __restrict__ witness, int bits,
                                                                const P2Consts* __restrict__ k, int pad) {
  __shared__ uint32_t found;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) found = witness[b];
  __syncthreads();
  for (uint32_t lo = kGrindRange * (uint32_t)kGrindLaunches; found == 0xffffffffu && lo < kP - kThreads; lo += kThreads) {
    const uint32_t w = lo + threadIdx.x;
    const bool ok = grind_try(ch + b, w, (1u << bits) - 1, k, pad);
    __syncthreads();
    if (ok) atomicMin(&found, w);
    __syncthreads();
  }
  if (threadIdx.x == 0) witness[b] = found;
}

__global__ __launch_bounds__(kChThreads) void ch_queries_kernel(DevChallenger* ch, const uint32_t* __restrict__ witness,
                                                               uint32_t* __restrict__ indices, int n_queries,
                                                               int pow_bits, int index_bits, int batch,
                                                               const P2Consts* __restrict__ k, int align) {
  const int t = blockIdx.x * kChThreads + threadIdx.x, b = t >> 4, e = t & 15;
  if (b >= batch) return;
  Ch c;
  ch_load(ch + b, c, e, k);
  ch_observe(c, Fp::from_canonical(witness[b]), k);
  if (align)
    while (c.n_in != 0) ch_observe(c, Fp::zero(), k);
  (void)ch_sample(c, k);  // the proof-of-work sample (zero in its low pow_bits by construction)
  if (align) c.n_out = 0;  // (format v16: the query indices start from a fresh squeeze)
  (void)pow_bits;
  for (int q = 0; q < n_queries; ++q) {
    uint32_t v = ch_sample(c, k).to_canonical() & ((1u << index_bits) - 1);
    if (e == 0) indices[(size_t)b * n_queries + q] = v;
  }
  ch_store(ch + b, c);
}

void launch_ch_init(hipStream_t stream, DevChallenger* ch, const uint32_t* init_obs, int n_obs, int batch,
                    const P2Consts* consts) {
  hipLaunchKernelGGL(ch_init_kernel, dim3((batch * 16 + kChThreads - 1) / kChThreads), dim3(kChThreads), 0, stream, ch, init_obs, n_obs, batch, consts);
}
void launch_ch_observe_sample(hipStream_t stream, DevChallenger* ch, const uint32_t* obs, size_t obs_stride, int n_obs,
                              uint32_t* out, size_t out_stride, int n_ext, int batch, const P2Consts* consts, int align) {
  hipLaunchKernelGGL(ch_observe_sample_kernel, dim3((batch * 16 + kChThreads - 1) / kChThreads), dim3(kChThreads), 0, stream, ch, obs, obs_stride,
                     n_obs, out, out_stride, n_ext, batch, consts, align);
}
void launch_ch_grind(hipStream_t stream, DevChallenger* ch, uint32_t* witness, int bits, int batch,
                     const P2Consts* consts, int pad) {
  (void)hipMemsetAsync(witness, 0xff, (size_t)batch * 4, stream);
  // The first kGrindRange * kGrindLaunches = 2^20 candidates are covered by launches of equal
  // ranges; a later launch returns at once for a proof whose witness is already known.  The
  // range grows as the batch shrinks (about 2^22 lanes per launch, at most 2^18 per proof): a
  // single proof needs 4 launches instead of 64, a batch of 256 keeps 64 launches of 2^14.
  // Every candidate below the witness found is tested either way, so the result is the
  // minimal witness regardless of the split.
  int log_range = 14;
  while (log_range < 18 && ((size_t)batch << (log_range + 1)) <= ((size_t)1 << 22)) ++log_range;
  const uint32_t range = 1u << log_range;
  const int launches = (int)((kGrindRange * (uint32_t)kGrindLaunches) >> log_range);
  for (int r = 0; r < launches; ++r)
    hipLaunchKernelGGL(ch_grind_kernel, dim3(range / kThreads, batch), dim3(kThreads), 0, stream, ch, witness, bits,
                       range * (uint32_t)r, consts, pad);
  hipLaunchKernelGGL(ch_grind_tail_kernel, dim3(batch), dim3(kThreads), 0, stream, ch, witness, bits, consts, pad);
}
void launch_ch_queries(hipStream_t stream, DevChallenger* ch, const uint32_t* witness, uint32_t* indices,
                       int n_queries, int pow_bits, int index_bits, int batch, const P2Consts* consts, int align) {
  hipLaunchKernelGGL(ch_queries_kernel, dim3((batch * 16 + kChThreads - 1) / kChThreads), dim3(kChThreads), 0, stream, ch, witness, indices, n_queries,
                     pow_bits, index_bits, batch, consts, align);
}

// Coefficients of the reduced openings (machine proofs, format v16; mverifier.cpp machine_reduce_exponents):
// out[b][t] = delta[b]^(desc[t] >> 16) * af[b]^(desc[t] & 0xffff), af = chal[b][0..4], delta = chal[b][4..8]
__global__ __launch_bounds__(kThreads) void reduce_coefs_kernel(const uint32_t* __restrict__ chal, size_t chal_stride,
                                                               const uint32_t* __restrict__ desc, uint32_t* __restrict__ out,
                                                               size_t out_stride, int n) {
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= n) return;
  const int b = blockIdx.y;
  const uint32_t d = desc[t];
  uint32_t e = d & 0xffffu;
  Fp4 x = load_fp4(chal + (size_t)b * chal_stride);
  const Fp4 dl = load_fp4(chal + (size_t)b * chal_stride + 4);
  Fp4 r = Fp4::one();
  for (uint32_t i = 0; i < (d >> 16); ++i) r = r * dl;
  while (e) {
    if (e & 1) r = r * x;
    x = x.sqr();
    e >>= 1;
  }
  store_fp4(out + (size_t)b * out_stride + (size_t)t * 4, r);
}
void launch_reduce_coefs(hipStream_t stream, const uint32_t* chal, size_t chal_stride, const uint32_t* desc, uint32_t* out,
                         size_t out_stride, int n, int batch) {
  hipLaunchKernelGGL(reduce_coefs_kernel, dim3((n + kThreads - 1) / kThreads, batch), dim3(kThreads), 0, stream, chal, chal_stride,
                     desc, out, out_stride, n);
}

#ifdef ZKSP_COMPONENT  // the round-1 keccak-chip component path only (include/zksp_component.h)
// ===========================================================================
// proof assembly: everything the verifier reads, canonical u32, one workgroup
// per query plus one for the fixed part
// ===========================================================================
__device__ __forceinline__ size_t layer_off(int logn, int layer) {
  // digests before `layer` in a tree with 2^logn leaves
  return ((size_t)2 << logn) - ((size_t)2 << (logn - layer));
}
__device__ __forceinline__ void put_path(uint32_t* dst, const uint32_t* tree, int logn, size_t idx) {
  // 8 * logn words, cooperatively
  for (int t = threadIdx.x; t < 8 * logn; t += kThreads) {
    int l = t >> 3, j = t & 7;
    size_t sib = (idx >> l) ^ 1;
    dst[t] = Fp::raw(tree[(layer_off(logn, l) + sib) * 8 + j]).to_canonical();
  }
}

__global__ __launch_bounds__(kThreads) void assemble_kernel(AssembleArgs a) {
  const int b = blockIdx.y, q = blockIdx.x;
  const int W = a.width, logh = a.logh, logn = logh + 1;
  const size_t h = (size_t)1 << logh, n = 2 * h;
  uint32_t* body = a.body + (size_t)b * a.body_stride;
  const uint32_t* tree_t = a.tree_t + (size_t)b * a.tree_t_stride;
  const uint32_t* tree_q = a.tree_q + (size_t)b * a.tree_q_stride;
  const uint32_t* tree_p = a.tree_p + (size_t)b * a.tree_p_stride;
  const uint32_t* fl = a.fri_layers + (size_t)b * a.fri_layer_stride;
  const uint32_t* ft = a.fri_trees + (size_t)b * a.fri_tree_stride;
  // body: trace root 8 | running-sum root 8 | cumulative sum 4 | quotient root 8 | opened | FRI roots | final | witness | queries
  const size_t n_open_words = (size_t)(2 * W + 8 + 2 * ka::kPermWidth) * 4;
  const size_t off_opened = 28, off_fri_roots = off_opened + n_open_words, off_final = off_fri_roots + 8 * (size_t)logh,
               off_witness = off_final + 4, off_queries = off_witness + 1;
  if (q == a.n_queries) {
    // fixed part
    for (int t = threadIdx.x; t < 8; t += kThreads) {
      body[t] = Fp::raw(tree_t[(2 * n - 2) * 8 + t]).to_canonical();
      body[8 + t] = Fp::raw(tree_p[(2 * n - 2) * 8 + t]).to_canonical();
      body[20 + t] = Fp::raw(tree_q[(2 * n - 2) * 8 + t]).to_canonical();
    }
    if (threadIdx.x < 4) body[16 + threadIdx.x] = Fp::raw(a.cum_sum[(size_t)b * 4 + threadIdx.x]).to_canonical();
    const uint32_t* op = a.opened + (size_t)b * a.opened_stride;
    for (size_t t = threadIdx.x; t < n_open_words; t += kThreads) body[off_opened + t] = Fp::raw(op[t]).to_canonical();
    size_t toff = 0, loff = 0;
    for (int k = 0; k < logh; ++k) {
      size_t hk = h >> k;
      if (threadIdx.x < 8)
        body[off_fri_roots + 8 * (size_t)k + threadIdx.x] = Fp::raw(ft[(toff + 2 * hk - 2) * 8 + threadIdx.x]).to_canonical();
      toff += 2 * hk - 1;
      loff += 2 * hk * 4;
    }
    if (threadIdx.x < 4) body[off_final + threadIdx.x] = Fp::raw(fl[loff + threadIdx.x]).to_canonical();
    if (threadIdx.x == 0) body[off_witness] = a.witness[b];
    return;
  }
  size_t perq = (size_t)W + 8 * (size_t)logn + ka::kPermWidth + 8 * (size_t)logn + 8 + 8 * (size_t)logn;
  for (int k = 0; k < logh; ++k) perq += 8 + 8 * (size_t)(logh - k);
  uint32_t* dst = body + off_queries + perq * (size_t)q;
  const size_t idx = a.indices[(size_t)b * a.n_queries + q];
  const size_t c = idx >> logh, m = idx & (h - 1);
  const uint32_t* lt = a.lde_t + (size_t)b * a.lde_t_stride + c * h + m;
  for (int i = threadIdx.x; i < W; i += kThreads) dst[i] = Fp::raw(lt[(size_t)i * n]).to_canonical();
  dst += W;
  put_path(dst, tree_t, logn, idx);
  dst += 8 * logn;
  const uint32_t* lp = a.lde_p + (size_t)b * a.lde_p_stride + c * h + m;
  if (threadIdx.x < ka::kPermWidth) dst[threadIdx.x] = Fp::raw(lp[(size_t)threadIdx.x * n]).to_canonical();
  dst += ka::kPermWidth;
  put_path(dst, tree_p, logn, idx);
  dst += 8 * logn;
  const uint32_t* lq = a.lde_q + (size_t)b * a.lde_q_stride + c * h + m;
  if (threadIdx.x < 8) dst[threadIdx.x] = Fp::raw(lq[(size_t)threadIdx.x * n]).to_canonical();
  dst += 8;
  put_path(dst, tree_q, logn, idx);
  dst += 8 * logn;
  size_t toff = 0, loff = 0;
  for (int k = 0; k < logh; ++k) {
    const int loghk = logh - k;
    const size_t hk = h >> k, half = hk >> 1;
    const size_t mk = m & (half - 1);
    const size_t leaf = c * half + mk;
    if (threadIdx.x < 4) {
      dst[threadIdx.x] = Fp::raw(fl[loff + (c * hk + mk) * 4 + threadIdx.x]).to_canonical();
      dst[4 + threadIdx.x] = Fp::raw(fl[loff + (c * hk + mk + half) * 4 + threadIdx.x]).to_canonical();
    }
    dst += 8;
    put_path(dst, ft + toff * 8, loghk, leaf);
    dst += 8 * loghk;
    toff += 2 * hk - 1;
    loff += 2 * hk * 4;
  }
}

void launch_assemble(hipStream_t stream, const AssembleArgs& a) {
  hipLaunchKernelGGL(assemble_kernel, dim3(a.n_queries + 1, a.batch), dim3(kThreads), 0, stream, a);
}

#endif  // ZKSP_COMPONENT

}  // namespace zksp
