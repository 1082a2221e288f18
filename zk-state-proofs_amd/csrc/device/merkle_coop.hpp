// Cooperative (16 lanes per permutation) Merkle pieces shared by the hashing kernels and the
// fused FRI tail: the top of a tree, and the whole commitment of a small FRI layer, done by ONE
// workgroup of kTopThreads threads.  Replaces the corresponding parts of p3-merkle-tree / p3-fri
// 0.1.4-succinct (reference Cargo.lock:5336, :5253).
#pragma once
#include "poseidon2_coop.hpp"

namespace zksp {

// These levels are a chain of dependent permutations with little width, so they use the
// cooperative 16-lane permutation: 64 compressions in flight per workgroup.
constexpr int kTopThreads = 1024;
constexpr int kTopGroups = kTopThreads / 16;

// Finishes a tree from a layer of `count` digests at t[in_off ..]; every thread of the
// workgroup must call it (it synchronises after each level).
__device__ __forceinline__ void coop_tree_levels(uint32_t* __restrict__ t, size_t in_off, int count,
                                                 const CoopConsts& cc, const P2Consts* __restrict__ consts) {
  const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
  while (count > 1) {
    const int parents = count >> 1;
    const size_t out_off = in_off + (size_t)count;
    for (int p0 = 0; p0 < parents; p0 += kTopGroups) {
      const int p = p0 + grp;
      const bool act = p < parents;
      // children 2p and 2p+1 are adjacent: lane e takes word e of the 16-word pair
      Fp x = act ? Fp::raw(t[(in_off + 2 * (size_t)p) * 8 + e]) : Fp::zero();
      x = p2_permute_coop(x, cc, consts);
      if (act && e < 8) t[(out_off + (size_t)p) * 8 + e] = x.v;
    }
    __syncthreads();
    in_off = out_off;
    count = parents;
  }
}

// The same for ONE SUBTREE of a wide layer: the workgroup takes digests [sub * sub_count, (sub + 1) * sub_count) of the
// layer of `count` digests at t[in_off ..] and climbs log2(sub_count) levels, writing its part of every layer on the way
// (layers are stored one after the other, so the next one starts `count` digests further on).  A single proof's tree of
// 2^17 leaves takes two launches this way instead of nine.
__device__ __forceinline__ void coop_subtree_levels(uint32_t* __restrict__ t, size_t in_off, int count, int sub, int sub_count,
                                                    const CoopConsts& cc, const P2Consts* __restrict__ consts) {
  const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
  size_t in_base = in_off + (size_t)sub * sub_count;
  while (sub_count > 1) {
    const int parents = sub_count >> 1;
    const size_t out_off = in_off + (size_t)count, out_base = out_off + (size_t)sub * parents;
    for (int p0 = 0; p0 < parents; p0 += kTopGroups) {
      const int p = p0 + grp;
      const bool act = p < parents;
      Fp x = act ? Fp::raw(t[(in_base + 2 * (size_t)p) * 8 + e]) : Fp::zero();
      x = p2_permute_coop(x, cc, consts);
      if (act && e < 8) t[(out_base + (size_t)p) * 8 + e] = x.v;
    }
    __syncthreads();
    in_off = out_off;
    in_base = out_base;
    count >>= 1;
    sub_count = parents;
  }
}

// Whole commitment of one FRI layer f = [2][hk] Fp4 into the tree t: leaf (c, m) =
// (f[c][m], f[c][m + hk/2]) absorbed into a zero state, then the levels.  Root at t[(2 hk - 2) * 8].
__device__ __forceinline__ void coop_fri_commit_block(const uint32_t* __restrict__ f, uint32_t* __restrict__ t,
                                                      int loghk, const CoopConsts& cc,
                                                      const P2Consts* __restrict__ consts) {
  const int hk = 1 << loghk, half = hk >> 1;
  const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
  for (int l0 = 0; l0 < hk; l0 += kTopGroups) {
    const int leaf = l0 + grp;
    const bool act = leaf < hk;
    const int c = leaf >= half ? 1 : 0, m = leaf - c * half;
    Fp x = Fp::zero();
    if (act && e < 8) x = Fp::raw(f[((size_t)c * hk + m + (e >= 4 ? half : 0)) * 4 + (e & 3)]);
    x = p2_permute_coop(x, cc, consts);
    if (act && e < 8) t[(size_t)leaf * 8 + e] = x.v;
  }
  __syncthreads();
  coop_tree_levels(t, 0, hk, cc, consts);
}

}  // namespace zksp
