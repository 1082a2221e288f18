// Poseidon2-BabyBear Merkle commitment (hot-path row a5; replaces p3-merkle-tree /
// p3-symmetric 0.1.4-succinct, reference Cargo.lock:5336, :5367, reached beneath
// prover/src/bin/main.rs:71-74).
//
// Leaf layer: one lane per matrix row, the 16-word sponge state lives in VGPRs.
// The matrix is column-major, so at every absorb step the 64 lanes of a wave read
// 64 consecutive words of one column: each column is streamed from HBM exactly
// once, fully coalesced.  Upper layers: one lane per parent; once a layer fits one
// workgroup the rest of the tree is finished in a single launch.
#include "kernels.h"
#include "merkle_coop.hpp"

namespace zksp {

constexpr int kHashThreads = 256;

__device__ __forceinline__ void leaf_hash_body(const uint32_t* __restrict__ mat, size_t mat_stride, int width,
                                               int n_rows, uint32_t* __restrict__ tree, size_t tree_stride,
                                               const P2Consts* __restrict__ consts) {
  const int row = blockIdx.x * kHashThreads + threadIdx.x;
  if (row >= n_rows) return;
  const uint32_t* m = mat + (size_t)blockIdx.y * mat_stride + row;
  Fp s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = Fp::zero();
  int c0 = 0;
  for (; c0 + 8 <= width; c0 += 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = Fp::raw(m[(size_t)(c0 + i) * n_rows]);
    p2_permute(s, consts);
  }
  if (c0 < width) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = c0 + i < width ? Fp::raw(m[(size_t)(c0 + i) * n_rows]) : Fp::zero();  // the last block is zero-filled
    p2_permute(s, consts);
  }
  uint4* d = reinterpret_cast<uint4*>(tree + (size_t)blockIdx.y * tree_stride + (size_t)row * 8);
  d[0] = make_uint4(s[0].v, s[1].v, s[2].v, s[3].v);
  d[1] = make_uint4(s[4].v, s[5].v, s[6].v, s[7].v);
}

// Two entry points over one body so that the wide trace commitment (the dominant
// kernel of a proof) carries its own name in rocprofv3 kernel statistics.
__global__ __launch_bounds__(kHashThreads) void leaf_hash_kernel(const uint32_t* __restrict__ mat, size_t mat_stride,
                                                                int width, int n_rows, uint32_t* __restrict__ tree,
                                                                size_t tree_stride,
                                                                const P2Consts* __restrict__ consts) {
  leaf_hash_body(mat, mat_stride, width, n_rows, tree, tree_stride, consts);
}
__global__ __launch_bounds__(kHashThreads) void leaf_hash_trace_kernel(const uint32_t* __restrict__ mat,
                                                                      size_t mat_stride, int width, int n_rows,
                                                                      uint32_t* __restrict__ tree, size_t tree_stride,
                                                                      const P2Consts* __restrict__ consts) {
  leaf_hash_body(mat, mat_stride, width, n_rows, tree, tree_stride, consts);
}

__device__ __forceinline__ void compress_pair(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                              const P2Consts* __restrict__ consts) {
  const uint4* p = reinterpret_cast<const uint4*>(src);
  uint4 a = p[0], b = p[1], c = p[2], d = p[3];
  Fp s[16] = {Fp::raw(a.x), Fp::raw(a.y), Fp::raw(a.z), Fp::raw(a.w), Fp::raw(b.x), Fp::raw(b.y),
              Fp::raw(b.z), Fp::raw(b.w), Fp::raw(c.x), Fp::raw(c.y), Fp::raw(c.z), Fp::raw(c.w),
              Fp::raw(d.x), Fp::raw(d.y), Fp::raw(d.z), Fp::raw(d.w)};
  p2_permute(s, consts);
  uint4* q = reinterpret_cast<uint4*>(dst);
  q[0] = make_uint4(s[0].v, s[1].v, s[2].v, s[3].v);
  q[1] = make_uint4(s[4].v, s[5].v, s[6].v, s[7].v);
}

// one layer: parents [count] from children [2*count]
__global__ __launch_bounds__(kHashThreads) void compress_layer_kernel(uint32_t* __restrict__ tree, size_t tree_stride,
                                                                     size_t in_off, size_t out_off, int count,
                                                                     const P2Consts* __restrict__ consts) {
  const int i = blockIdx.x * kHashThreads + threadIdx.x;
  if (i >= count) return;
  uint32_t* t = tree + (size_t)blockIdx.y * tree_stride;
  compress_pair(t + (in_off + 2 * (size_t)i) * 8, t + (out_off + (size_t)i) * 8, consts);
}

// Finishes a tree from a layer of `count` (<= 2*kHashThreads) digests, one workgroup per proof
// (merkle_coop.hpp).
__global__ __launch_bounds__(kTopThreads) void compress_top_kernel(uint32_t* __restrict__ tree, size_t tree_stride,
                                                                  size_t in_off, int count,
                                                                  const P2Consts* __restrict__ consts) {
  const CoopConsts cc = coop_load_consts(consts, threadIdx.x & 15);
  coop_tree_levels(tree + (size_t)blockIdx.x * tree_stride, in_off, count, cc, consts);
}

// Whole commitment of a small FRI layer (<= 2*kHashThreads leaves) in one launch:
// leaf (c, m) = (f[c][m], f[c][m + Hk/2]) absorbed into a zero state, then the tree.
__global__ __launch_bounds__(kTopThreads) void fri_commit_small_kernel(const uint32_t* __restrict__ layer,
                                                                      size_t layer_stride, int loghk,
                                                                      uint32_t* __restrict__ tree, size_t tree_stride,
                                                                      const P2Consts* __restrict__ consts) {
  const CoopConsts cc = coop_load_consts(consts, threadIdx.x & 15);
  coop_fri_commit_block(layer + (size_t)blockIdx.x * layer_stride, tree + (size_t)blockIdx.x * tree_stride, loghk, cc,
                        consts);
}

// Latency form of the leaf layer for small batches: 16 lanes per matrix row, four
// rows per wave.  With few rows the lane-per-row kernel leaves most SIMDs idle and
// each lane walks its ceil(W/8) dependent permutations at full latency; here the
// same rows spread over 16x more lanes and every permutation is about 5x shorter.
// About 2x more VALU work in total, so it only pays while the chip is not full.
__global__ __launch_bounds__(kHashThreads) void leaf_hash_coop_kernel(const uint32_t* __restrict__ mat,
                                                                     size_t mat_stride, int width, int n_rows,
                                                                     uint32_t* __restrict__ tree, size_t tree_stride,
                                                                     const P2Consts* __restrict__ consts) {
  const int e = threadIdx.x & 15;
  const int row = blockIdx.x * (kHashThreads / 16) + (threadIdx.x >> 4);
  const bool act = row < n_rows;
  const CoopConsts cc = coop_load_consts(consts, e);
  const uint32_t* m = mat + (size_t)blockIdx.y * mat_stride + (act ? row : 0);
  // The sponge is a chain of ceil(width / 8) dependent permutations, but the words it absorbs
  // do not depend on it: they are fetched kPrefetch absorb steps ahead so that no step waits
  // for a global load (without this every step paid a full memory latency, 2.8 us of 2.8).
  constexpr int kPrefetch = 4;
  auto fetch = [&](int c) -> uint32_t {
    return (e < 8 && c + e < width) ? m[(size_t)(c + e) * n_rows] : 0u;
  };
  uint32_t ahead[kPrefetch];
#pragma unroll
  for (int j = 0; j < kPrefetch; ++j) ahead[j] = fetch(8 * j);
  int32_t x = 0;  // signed lazy word between permutations (poseidon2_coop.hpp)
  for (int c0 = 0; c0 < width; c0 += 8 * kPrefetch) {
    uint32_t cur[kPrefetch];
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j) cur[j] = ahead[j];
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j) ahead[j] = fetch(c0 + 8 * (kPrefetch + j));
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j) {
      const int c = c0 + 8 * j;
      if (c < width) {  // uniform
        if (e < 8) x = (int32_t)cur[j];  // (fetch() returns zero beyond the width: the last block is zero-filled)
        x = p2_permute_coop_signed(x, cc, consts);
      }
    }
  }
  if (act && e < 8) tree[(size_t)blockIdx.y * tree_stride + (size_t)row * 8 + e] = fps_canon(x);
}

// One subtree of kSubtree digests per workgroup, eight levels in one launch (merkle_coop.hpp).
constexpr int kSubtreeLog = 8, kSubtree = 1 << kSubtreeLog;
__global__ __launch_bounds__(kTopThreads) void compress_subtree_kernel(uint32_t* __restrict__ tree, size_t tree_stride,
                                                                      size_t in_off, int count,
                                                                      const P2Consts* __restrict__ consts) {
  const CoopConsts cc = coop_load_consts(consts, threadIdx.x & 15);
  coop_subtree_levels(tree + (size_t)blockIdx.y * tree_stride, in_off, count, (int)blockIdx.x, kSubtree, cc, consts);
}

// (from the layer of `count` digests that starts `off` digests into the tree)
static void launch_upper_layers_from(hipStream_t stream, size_t off, int count, uint32_t* tree, size_t tree_stride, int batch,
                                     const P2Consts* consts);
static void launch_upper_layers(hipStream_t stream, int logn, uint32_t* tree, size_t tree_stride, int batch,
                                const P2Consts* consts) {
  launch_upper_layers_from(stream, 0, 1 << logn, tree, tree_stride, batch, consts);
}
static void launch_upper_layers_from(hipStream_t stream, size_t off, int count, uint32_t* tree, size_t tree_stride, int batch,
                                     const P2Consts* consts) {
  // A batch of a few proofs: the levels are a chain of launches that each leave most of the GPU idle - climb eight levels
  // per launch, a subtree per workgroup, while the layer is wide enough (the cooperative permutation does twice the
  // arithmetic, which a small batch does not feel).
  while (batch <= 8 && count >= 4 * kSubtree) {
    hipLaunchKernelGGL(compress_subtree_kernel, dim3(count / kSubtree, batch), dim3(kTopThreads), 0, stream, tree, tree_stride, off,
                       count, consts);
    for (int l = 0; l < kSubtreeLog; ++l) {
      off += (size_t)count;
      count >>= 1;
    }
  }
  // One lane per parent while a level is wide: always above 512 digests, and in a large batch for
  // as long as the level has 16 K parents batch-wide (the cooperative form below does about twice
  // the arithmetic; it is for the narrow, latency-bound levels).
  while (count > 2 * kHashThreads || (count > 32 && (size_t)(count >> 1) * (size_t)batch >= 16384)) {
    int parents = count >> 1;
    hipLaunchKernelGGL(compress_layer_kernel, dim3((parents + kHashThreads - 1) / kHashThreads, batch),
                       dim3(kHashThreads), 0, stream, tree, tree_stride, off, off + (size_t)count, parents, consts);
    off += (size_t)count;
    count = parents;
  }
  if (count > 1)
    hipLaunchKernelGGL(compress_top_kernel, dim3(batch), dim3(kTopThreads), 0, stream, tree, tree_stride, off, count,
                       consts);
}

void launch_merkle_upper(hipStream_t stream, int logn, uint32_t* tree, size_t tree_stride, int batch,
                         const P2Consts* consts) {
  launch_upper_layers(stream, logn, tree, tree_stride, batch, consts);
}

void launch_merkle_commit(hipStream_t stream, const uint32_t* mat, size_t mat_stride, int width, int logn,
                          uint32_t* tree, size_t tree_stride, int batch, const P2Consts* consts, bool upper) {
  const int n = 1 << logn;
  const dim3 grid((n + kHashThreads - 1) / kHashThreads, batch);
  // fewer than 8 waves per CU of lane-per-row work: take the latency form
  if (width >= 64 && (size_t)n * (size_t)batch <= 32768) {
    const int rows_per_block = kHashThreads / 16;
    hipLaunchKernelGGL(leaf_hash_coop_kernel, dim3((n + rows_per_block - 1) / rows_per_block, batch),
                       dim3(kHashThreads), 0, stream, mat, mat_stride, width, n, tree, tree_stride, consts);
  } else if (width >= 64)
    hipLaunchKernelGGL(leaf_hash_trace_kernel, grid, dim3(kHashThreads), 0, stream, mat, mat_stride, width, n, tree,
                       tree_stride, consts);
  else
    hipLaunchKernelGGL(leaf_hash_kernel, grid, dim3(kHashThreads), 0, stream, mat, mat_stride, width, n, tree,
                       tree_stride, consts);
  if (upper) launch_upper_layers(stream, logn, tree, tree_stride, batch, consts);
}

// FRI layer commitment: leaf (c, m) = (f[c][m], f[c][m + Hk/2]), 8 words = one absorb
__global__ __launch_bounds__(kHashThreads) void fri_leaf_kernel(const uint32_t* __restrict__ layer,
                                                               size_t layer_stride, int loghk,
                                                               uint32_t* __restrict__ tree, size_t tree_stride,
                                                               const P2Consts* __restrict__ consts) {
  const int hk = 1 << loghk, half = hk >> 1;
  const int leaf = blockIdx.x * kHashThreads + threadIdx.x;
  if (leaf >= hk) return;
  const int c = leaf >= half ? 1 : 0, m = leaf - c * half;
  const uint4* f = reinterpret_cast<const uint4*>(layer + (size_t)blockIdx.y * layer_stride);
  uint4 lo = f[(size_t)c * hk + m], hi = f[(size_t)c * hk + m + half];
  Fp s[16];
  s[0] = Fp::raw(lo.x); s[1] = Fp::raw(lo.y); s[2] = Fp::raw(lo.z); s[3] = Fp::raw(lo.w);
  s[4] = Fp::raw(hi.x); s[5] = Fp::raw(hi.y); s[6] = Fp::raw(hi.z); s[7] = Fp::raw(hi.w);
#pragma unroll
  for (int i = 8; i < 16; ++i) s[i] = Fp::zero();
  p2_permute(s, consts);
  uint4* d = reinterpret_cast<uint4*>(tree + (size_t)blockIdx.y * tree_stride + (size_t)leaf * 8);
  d[0] = make_uint4(s[0].v, s[1].v, s[2].v, s[3].v);
  d[1] = make_uint4(s[4].v, s[5].v, s[6].v, s[7].v);
}

// A small batch's wide layer: a workgroup hashes the 256 leaves of its subtree (cooperative form) and climbs the eight
// levels above them in the same launch (fri_leaf_kernel + compress_subtree_kernel, one launch less per layer).
__global__ __launch_bounds__(kTopThreads) void fri_subtree_kernel(const uint32_t* __restrict__ layer, size_t layer_stride, int loghk,
                                                                 uint32_t* __restrict__ tree, size_t tree_stride,
                                                                 const P2Consts* __restrict__ consts) {
  const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const CoopConsts cc = coop_load_consts(consts, e);
  const uint32_t* f = layer + (size_t)blockIdx.y * layer_stride;
  uint32_t* t = tree + (size_t)blockIdx.y * tree_stride;
  const int hk = 1 << loghk, half = hk >> 1, leaf0 = (int)blockIdx.x * kSubtree;
  for (int l0 = 0; l0 < kSubtree; l0 += kTopGroups) {
    const int leaf = leaf0 + l0 + grp;  // (hk is a multiple of kSubtree here)
    const int c = leaf >= half ? 1 : 0, m = leaf - c * half;
    Fp x = Fp::zero();
    if (e < 8) x = Fp::raw(f[((size_t)c * hk + m + (e >= 4 ? half : 0)) * 4 + (e & 3)]);
    x = p2_permute_coop(x, cc, consts);
    if (e < 8) t[(size_t)leaf * 8 + e] = x.v;
  }
  __threadfence_block();
  __syncthreads();
  coop_subtree_levels(t, 0, hk, (int)blockIdx.x, kSubtree, cc, consts);
}

void launch_fri_commit(hipStream_t stream, const uint32_t* layer, size_t layer_stride, int loghk, uint32_t* tree,
                       size_t tree_stride, int batch, const P2Consts* consts) {
  const int hk = 1 << loghk;
  if (batch <= 8 && hk >= 4 * kSubtree) {
    hipLaunchKernelGGL(fri_subtree_kernel, dim3(hk / kSubtree, batch), dim3(kTopThreads), 0, stream, layer, layer_stride, loghk, tree,
                       tree_stride, consts);
    size_t off = 0;
    int count = hk;
    for (int l = 0; l < kSubtreeLog; ++l) {
      off += (size_t)count;
      count >>= 1;
    }
    launch_upper_layers_from(stream, off, count, tree, tree_stride, batch, consts);
    return;
  }
  // small layer: one cooperative workgroup per proof, unless the batch makes it wide (then one lane
  // per leaf / per parent does half the arithmetic and fills the chip)
  if (hk <= 2 * kHashThreads && (size_t)hk * (size_t)batch < 32768) {
    hipLaunchKernelGGL(fri_commit_small_kernel, dim3(batch), dim3(kTopThreads), 0, stream, layer, layer_stride, loghk,
                       tree, tree_stride, consts);
    return;
  }
  hipLaunchKernelGGL(fri_leaf_kernel, dim3((hk + kHashThreads - 1) / kHashThreads, batch), dim3(kHashThreads), 0,
                     stream, layer, layer_stride, loghk, tree, tree_stride, consts);
  launch_upper_layers(stream, loghk, tree, tree_stride, batch, consts);
}

__global__ __launch_bounds__(kHashThreads) void permute_kernel(uint32_t* __restrict__ states, size_t n,
                                                              const P2Consts* __restrict__ consts) {
  size_t i = (size_t)blockIdx.x * kHashThreads + threadIdx.x;
  if (i >= n) return;
  Fp s[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) s[k] = Fp::raw(states[i * 16 + k]);
  p2_permute(s, consts);
#pragma unroll
  for (int k = 0; k < 16; ++k) states[i * 16 + k] = s[k].v;
}

void launch_poseidon2_permute(hipStream_t stream, uint32_t* states, size_t n, const P2Consts* consts) {
  if (!n) return;
  hipLaunchKernelGGL(permute_kernel, dim3((unsigned)((n + kHashThreads - 1) / kHashThreads)), dim3(kHashThreads), 0,
                     stream, states, n, consts);
}

}  // namespace zksp
