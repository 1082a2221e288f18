// Host-side Poseidon2 sponge, compression, duplex challenger and Merkle helpers shared by the two
// verifiers (verifier.cpp: keccak-chip proofs, mverifier.cpp: machine proofs).  Same definitions the
// device uses (poseidon2.hpp), instantiated on the host.
#pragma once
#include <cstddef>
#include <vector>

#include "../device/poseidon2.hpp"

namespace zksp {
namespace hosthash {

struct HostChallenger {
  Fp state[16];
  Fp inbuf[8];
  Fp outbuf[8];
  int n_in = 0, n_out = 0;
  const P2Consts* k;
  // optional record of every duplex: 1 if it absorbed a block (0: a squeeze), then the 16 words the permutation started from
  // (machine proofs since format v16 end every phase on a block boundary, so a duplex absorbs eight words or none)
  std::vector<uint32_t>* record = nullptr;
  explicit HostChallenger(const P2Consts* kk) : k(kk) {
    for (auto& s : state) s = Fp::zero();
  }
  void duplex() {
    for (int i = 0; i < n_in; ++i) state[i] = inbuf[i];
    if (record) {
      record->push_back(n_in == 8 ? 1u : n_in == 0 ? 0u : 2u);  // (2: a partial block - not a machine-proof transcript)
      for (int i = 0; i < 16; ++i) record->push_back(state[i].to_canonical());
    }
    n_in = 0;
    p2_permute(state, k);
    for (int i = 0; i < 8; ++i) outbuf[i] = state[i];
    n_out = 8;
  }
  void observe(Fp x) {
    n_out = 0;
    inbuf[n_in++] = x;
    if (n_in == 8) duplex();
  }
  void observe_canon(uint32_t c) { observe(Fp::from_canonical(c)); }
  Fp sample() {
    if (n_in != 0 || n_out == 0) duplex();
    return outbuf[--n_out];
  }
  Fp4 sample_ext() {
    Fp4 r;
    for (int i = 0; i < 4; ++i) r.c[i] = sample();
    return r;
  }
  uint32_t sample_bits(int bits) { return sample().to_canonical() & ((1u << bits) - 1); }
  // machine proofs since format v16: every phase of the transcript ends on a block boundary (a pending block is zero-filled),
  // so that a duplex is always "absorb eight words" or "squeeze" - the two row kinds of the in-circuit transcript
  void pad() {
    while (n_in != 0) observe(Fp::zero());
  }
  void drop_outputs() { n_out = 0; }  // the next sample starts from a fresh squeeze
};

inline void hash_elems(const Fp* in, size_t n, Fp out[8], const P2Consts* k) {
  Fp st[16];
  for (auto& s : st) s = Fp::zero();
  for (size_t off = 0; off < n; off += 8) {
    size_t m = n - off < 8 ? n - off : 8;
    for (size_t i = 0; i < 8; ++i) st[i] = i < m ? in[off + i] : Fp::zero();  // overwrite mode; the last block is zero-filled
    p2_permute(st, k);
  }
  for (int i = 0; i < 8; ++i) out[i] = st[i];
}

inline void compress(const Fp* l, const Fp* r, Fp out[8], const P2Consts* k) {
  Fp st[16];
  for (int i = 0; i < 8; ++i) {
    st[i] = l[i];
    st[8 + i] = r[i];
  }
  p2_permute(st, k);
  for (int i = 0; i < 8; ++i) out[i] = st[i];
}

inline bool verify_path(const Fp leaf[8], size_t idx, const uint32_t* path_canon, int depth, const Fp root[8],
                 const P2Consts* k) {
  Fp cur[8];
  for (int i = 0; i < 8; ++i) cur[i] = leaf[i];
  for (int l = 0; l < depth; ++l) {
    Fp sib[8], nxt[8];
    for (int i = 0; i < 8; ++i) sib[i] = Fp::from_canonical(path_canon[8 * l + i]);
    if ((idx >> l) & 1) compress(sib, cur, nxt, k);
    else compress(cur, sib, nxt, k);
    for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
  }
  for (int i = 0; i < 8; ++i)
    if (cur[i] != root[i]) return false;
  return true;
}

// Merkle root of a flat word list laid out column-major as [8][2^logr], zero padded: the
// form in which long lists (opened values, the public I/O limbs) enter the transcript.
inline void list_root(const std::vector<Fp>& words, int logr, Fp out[8], const P2Consts* kc) {
  const size_t R = (size_t)1 << logr;
  std::vector<Fp> pad(8 * R, Fp::zero());
  for (size_t t = 0; t < words.size() && t < 8 * R; ++t) pad[t] = words[t];
  std::vector<Fp> layer(8 * R), nxt;
  for (size_t r = 0; r < R; ++r) {
    Fp row[8];
    for (int c = 0; c < 8; ++c) row[c] = pad[(size_t)c * R + r];
    hash_elems(row, 8, &layer[8 * r], kc);
  }
  for (size_t cnt = R; cnt > 1; cnt >>= 1) {
    nxt.assign(8 * (cnt / 2), Fp::zero());
    for (size_t i = 0; i < cnt / 2; ++i) compress(&layer[16 * i], &layer[16 * i + 8], &nxt[8 * i], kc);
    layer.swap(nxt);
  }
  for (int i = 0; i < 8; ++i) out[i] = layer[i];
}

}  // namespace hosthash
}  // namespace zksp
