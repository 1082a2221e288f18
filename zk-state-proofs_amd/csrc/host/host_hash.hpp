// Host-side Poseidon2 sponge, compression, duplex challenger and Merkle helpers shared by the two
// verifiers (verifier.cpp: keccak-chip proofs, mverifier.cpp: machine proofs).  Same definitions the
// device uses (poseidon2.hpp), instantiated on the host.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <atomic>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../device/poseidon2.hpp"

namespace zksp {
namespace p2avx2 {  // p2_avx2.cpp: the permutation on the host's vector unit (same function, bit-identical results)
void permute(uint32_t* s, const uint32_t (*ext)[16], const uint32_t* internal, const uint32_t* diag);
void permute2(uint32_t* sa, uint32_t* sb, const uint32_t (*ext)[16], const uint32_t* internal, const uint32_t* diag);  // two states in lockstep
bool usable();
}  // namespace p2avx2
namespace p2avx512 {  // p2_avx512.cpp: a state per 512-bit register, up to four states in lockstep
void permute1(uint32_t* a, const uint32_t (*ext)[16], const uint32_t* internal, const uint32_t* diag);
void permute2(uint32_t* a, uint32_t* b, const uint32_t (*ext)[16], const uint32_t* internal, const uint32_t* diag);
void permute4(uint32_t* a, uint32_t* b, uint32_t* c, uint32_t* d, const uint32_t (*ext)[16], const uint32_t* internal, const uint32_t* diag);
bool usable();
}  // namespace p2avx512
namespace hosthash {

// Which form of the permutation this host runs: 2 = AVX-512 (a state per register, four states in lockstep), 1 = AVX2 (two
// registers per state, two states in lockstep), 0 = the device's signed lazy form on a core.  ZKSP_HOST_P2=scalar|avx2 lowers it
// (measurements, and the tests' comparison of the forms through the whole verifier).
inline int vector_level() {
  static const int level = [] {
    int l = p2avx512::usable() ? 2 : p2avx2::usable() ? 1 : 0;
    if (const char* e = getenv("ZKSP_HOST_P2")) {
      const std::string v(e);
      if (v == "scalar") l = 0;
      else if (v == "avx2") l = l < 1 ? l : 1;
    }
    return l;
  }();
  return level;
}
// how many queries the verifier checks in lockstep, so that their permutations fill the vector unit
inline int lockstep_lanes() { return vector_level() == 2 ? 4 : 2; }

// The host's permutation.  State: Montgomery words in [0, p).
inline void permute(Fp* st, const P2Consts* k) {
  const int level = vector_level();
  if (level == 2) p2avx512::permute1(&st[0].v, k->ext, k->internal, k->diag);
  else if (level == 1) p2avx2::permute(&st[0].v, k->ext, k->internal, k->diag);
  else p2_permute(st, k);
}
// L independent states in lockstep where a vector form exists (one permutation is a chain of dependent operations; further
// chains beside it cost a fraction of the time each)
template <int L>
inline void permute_lanes(Fp (*st)[16], const P2Consts* k) {
  const int level = vector_level();
  int t = 0;
  if (level == 2) {
    for (; t + 4 <= L; t += 4) p2avx512::permute4(&st[t][0].v, &st[t + 1][0].v, &st[t + 2][0].v, &st[t + 3][0].v, k->ext, k->internal, k->diag);
    for (; t + 2 <= L; t += 2) p2avx512::permute2(&st[t][0].v, &st[t + 1][0].v, k->ext, k->internal, k->diag);
  } else if (level == 1) {
    for (; t + 2 <= L; t += 2) p2avx2::permute2(&st[t][0].v, &st[t + 1][0].v, k->ext, k->internal, k->diag);
  }
  for (; t < L; ++t) permute(st[t], k);
}

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
    permute(state, k);
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
    permute(st, k);
  }
  for (int i = 0; i < 8; ++i) out[i] = st[i];
}

inline void compress(const Fp* l, const Fp* r, Fp out[8], const P2Consts* k) {
  Fp st[16];
  for (int i = 0; i < 8; ++i) {
    st[i] = l[i];
    st[8 + i] = r[i];
  }
  permute(st, k);
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
// n_threads > 1: the subtrees below the top levels are hashed side by side (a long list is 2^12 rows: 8 000 permutations).
inline void list_root(const std::vector<Fp>& words, int logr, Fp out[8], const P2Consts* kc, unsigned n_threads = 1) {
  const size_t R = (size_t)1 << logr;
  auto word = [&](size_t t) { return t < words.size() ? words[t] : Fp::zero(); };
  // root of the subtree over rows [r0, r0 + cnt)
  auto subtree = [&](size_t r0, size_t cnt, Fp* root) {
    std::vector<Fp> layer(8 * cnt);
    for (size_t r = 0; r < cnt; ++r) {
      Fp row[8];
      for (int c = 0; c < 8; ++c) row[c] = word((size_t)c * R + r0 + r);
      hash_elems(row, 8, &layer[8 * r], kc);
    }
    for (size_t n = cnt; n > 1; n >>= 1)
      for (size_t i = 0; i < n / 2; ++i) {  // (in place: node i of the next level overwrites slot i, which is already consumed)
        Fp t[8];
        compress(&layer[16 * i], &layer[16 * i + 8], t, kc);
        for (int k = 0; k < 8; ++k) layer[8 * i + k] = t[k];
      }
    for (int i = 0; i < 8; ++i) root[i] = layer[i];
  };
  int lt = 0;
  while (((size_t)2 << lt) <= n_threads && lt + 6 <= logr) ++lt;  // 2^lt subtrees of at least 64 rows each
  const size_t parts = (size_t)1 << lt;
  std::vector<Fp> top(8 * parts);
  if (parts == 1) {
    subtree(0, R, top.data());
  } else {
    struct Joiner {
      std::vector<std::thread> th;
      ~Joiner() { for (auto& t : th) if (t.joinable()) t.join(); }
    } pool;
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};  // (a part that ran out of memory: reported by the calling thread, never thrown on another)
    auto worker = [&]() noexcept {
      for (size_t p; (p = next.fetch_add(1)) < parts;) {
        try {
          subtree(p * (R / parts), R / parts, &top[8 * p]);
        } catch (...) {
          failed = true;
        }
      }
    };
    {
      Joiner& jp = pool;
      try {
        for (size_t t = 1; t < parts; ++t) jp.th.emplace_back(worker);
      } catch (...) {
      }  // fewer threads than wanted: the parts are claimed from one counter
      worker();
      for (auto& t : jp.th) if (t.joinable()) t.join();
    }
    if (failed) throw std::bad_alloc();
  }
  for (size_t n = parts; n > 1; n >>= 1)
    for (size_t i = 0; i < n / 2; ++i) {
      Fp t[8];
      compress(&top[16 * i], &top[16 * i + 8], t, kc);
      for (int k = 0; k < 8; ++k) top[8 * i + k] = t[k];
    }
  for (int i = 0; i < 8; ++i) out[i] = top[i];
}

}  // namespace hosthash
}  // namespace zksp
