// Host verifier of the machine proof ("ZKSP v15"): replaces `client.verify(&proof, &vk)` (reference
// prover/src/bin/main.rs:80; sp1-stark 3.4.0's multi-chip verifier over p3-uni-stark / p3-fri,
// Cargo.lock:7485, :5378, :5253) for proofs that bind the guest's whole execution.  Also the
// host half of `client.setup(ELF)` (main.rs:70): the commitment to the preprocessed Program and
// Image tables that the verifying key carries.  Needs no GPU.
//
// Checks, in order: header sanity and key; sha256(public values) = committed digest; exit code 0;
// the LogUp buses balance (chips' cumulative sums + the public COMMIT / HALT terms = 0); for every
// chip the constraint identity at zeta with the SAME templates the device quotient kernels
// instantiate (air_machine.hpp, air_keccak.hpp); proof of work; every FRI query: the four
// mixed-height Merkle openings, the reduced openings per height, the folding chain.
#include "mverifier.hpp"

#include <atomic>
#include <thread>

#include <array>
#include <type_traits>
#include <cstring>

#include "host_hash.hpp"
#include "zeta_program.hpp"
#include "machine_defs.hpp"

namespace zksp {

using namespace hosthash;
using namespace mach;

namespace {

inline uint32_t bitrev32(uint32_t v, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}
int ceil_log2(size_t v) {
  int l = 0;
  while (((size_t)1 << l) < v) ++l;
  return l;
}
Fp4 read_fp4(const uint32_t* w) {
  Fp4 r;
  for (int i = 0; i < 4; ++i) r.c[i] = Fp::from_canonical(w[i]);
  return r;
}
Fp4 from_basis(const Fp4* v) {  // four opened base columns = one extension column
  Fp4 r = Fp4::zero();
  for (int j = 0; j < 4; ++j) {
    Fp4 basis = Fp4::zero();
    basis.c[j] = Fp::one();
    r += basis * v[j];
  }
  return r;
}

// ---- host NTT / LDE for the preprocessed tables (setup only) ----
void host_ntt(std::vector<Fp>& a, int logn, bool inverse) {
  const size_t n = (size_t)1 << logn;
  for (size_t i = 0; i < n; ++i) {
    size_t j = bitrev32((uint32_t)i, logn);
    if (i < j) std::swap(a[i], a[j]);
  }
  for (int s = 1; s <= logn; ++s) {
    const size_t m = (size_t)1 << s, half = m >> 1;
    Fp wm = fp_root_of_unity(s);
    if (inverse) wm = wm.inv();
    for (size_t k = 0; k < n; k += m) {
      Fp w = Fp::one();
      for (size_t j = 0; j < half; ++j) {
        const Fp t = w * a[k + j + half], u = a[k + j];
        a[k + j] = u + t;
        a[k + j + half] = u - t;
        w = w * wm;
      }
    }
  }
  if (inverse) {
    const Fp ninv = Fp::from_canonical((uint32_t)(n % kP)).inv();
    for (auto& x : a) x = x * ninv;
  }
}
// column of H evaluations -> [2][H] evaluations over g*K_H and g*w_2H*K_H
void host_lde(const std::vector<Fp>& col, int logh, std::vector<Fp>* out) {
  const size_t h = (size_t)1 << logh;
  std::vector<Fp> c = col;
  host_ntt(c, logh, true);
  out->assign(2 * h, Fp::zero());
  const Fp g = Fp::from_canonical(kGen), w2h = fp_root_of_unity(logh + 1);
  for (int cs = 0; cs < 2; ++cs) {
    const Fp shift = cs ? g * w2h : g;
    std::vector<Fp> t(h);
    Fp p = Fp::one();
    for (size_t k = 0; k < h; ++k) { t[k] = c[k] * p; p = p * shift; }
    host_ntt(t, logh, false);
    for (size_t k = 0; k < h; ++k) (*out)[(size_t)cs * h + k] = t[k];
  }
}

struct RoundShape {
  int width[kNumChips];
  int lm;  // tallest log height in the round
};

// ---- leaf-check log (mverifier.hpp LeafCheckLog): the permutations of the query phase as Poseidon2-chip row records ----
// (32 words: flags, tag, key, mask, the 16 input words, root id, the Horner sum after the block, alpha_f, padding)
// alpha_f as the sponge rows use it: its powers for Horner's rule over a block of eight words, and its canonical words
struct AlphaPows {
  Fp4 pw[8];  // alpha^7 .. alpha^0
  Fp4 a8;
  uint32_t canon[4];
  explicit AlphaPows(const Fp4& a) {
    pw[7] = Fp4::one();
    for (int i = 6; i >= 0; --i) pw[i] = pw[i + 1] * a;
    a8 = pw[0] * a;
    for (int i = 0; i < 4; ++i) canon[i] = a.c[i].to_canonical();
  }
};
void log_p2_row(LeafCheckLog* log, uint32_t flags, uint32_t tag, uint32_t key, uint32_t mask, const Fp in[16], uint32_t rid = 0,
                const Fp4* so = nullptr, const AlphaPows* alpha = nullptr) {
  RowWords& v = log->p2_rows;
  const size_t at = v.size();
  v.resize(at + kP2RecWords);  // (not zero-filled)
  uint32_t* w = v.data() + at;
  w[0] = flags; w[1] = tag; w[2] = key; w[3] = mask;
  for (int i = 0; i < 16; ++i) w[4 + i] = in[i].to_canonical();
  w[kP2RecRid] = rid;
  for (int i = 0; i < 4; ++i) w[kP2RecSo + i] = so ? so->c[i].to_canonical() : 0u;
  for (int i = 0; i < 4; ++i) w[kP2RecAlpha + i] = alpha ? alpha->canon[i] : 0u;
  for (int i = kP2RecAlpha + 4; i < kP2RecWords; ++i) w[i] = 0u;
}
void log_pub_tuple(LeafCheckLog* log, uint32_t bus, bool verifier_sends, uint32_t mult, const uint32_t* el, int n_el) {
  std::vector<uint32_t>& v = log->pub_tuples;
  v.push_back(bus); v.push_back(verifier_sends ? 1u : 0u); v.push_back(mult); v.push_back((uint32_t)n_el);
  for (int i = 0; i < kPubTupleWords - 4; ++i) v.push_back(i < n_el ? el[i] : 0u);
}
// The hashing of the query phase works on L = 1 or 2 QUERIES in lockstep: two queries of one proof open rows of the same
// widths along paths of the same lengths, so their permutations pair up one to one, and the host's vector permutation runs two
// independent states almost as fast as one (host_hash.hpp permute_lanes).  Everything below takes per-lane arrays.
//
// hash_elems, every block logged as a sponge row labelled (tag, key, mask): the first starts from the zero state.  With
// `alpha` (the hash of a matrix row of a commitment): Horner's rule in alpha over the absorbed words, block by block; the last
// row is flagged SE and the sum comes back in *sum.
template <int L>
void sponge_logged(const Fp* const* in, size_t n, Fp (*out)[8], const P2Consts* kc, LeafCheckLog* const* log, const uint32_t* tag,
                   const uint32_t* key, uint32_t mask, bool run_start, bool send, bool fri_leaf, const AlphaPows* alpha = nullptr,
                   Fp4* const* sum = nullptr) {
  Fp st[L][16];
  Fp4 so[L];
  for (int t = 0; t < L; ++t) {
    for (auto& x : st[t]) x = Fp::zero();
    so[t] = Fp4::zero();
  }
  for (size_t off = 0; off < n; off += 8) {
    const size_t m = n - off < 8 ? n - off : 8;
    for (int t = 0; t < L; ++t) {
      for (size_t i = 0; i < 8; ++i) st[t][i] = i < m ? in[t][off + i] : Fp::zero();
      if (alpha) {
        Fp4 bv = Fp4::from_base(st[t][7]);
        for (int i = 0; i < 7; ++i) bv += alpha->pw[i] * st[t][i];
        so[t] = so[t] * alpha->a8 + bv;
      } else {
        so[t] = Fp4::from_base(st[t][7]);  // (a hash nobody reduces - a FRI pair: the row's alpha_f columns are zero, Horner's rule leaves the last word)
      }
      if (log[t]) {
        uint32_t flags = off == 0 ? (uint32_t)P2K_SZ : (uint32_t)P2K_SC;
        if (run_start) flags |= kP2FlagNew;
        if (send && off + 8 >= n) flags |= kP2FlagSnd;
        if (fri_leaf && off == 0) flags |= kP2FlagFri;
        if (alpha && off + 8 >= n) flags |= kP2FlagSe;
        log_p2_row(log[t], flags, tag[t], key[t], mask, st[t], 0, &so[t], alpha);
      }
    }
    permute_lanes<L>(st, kc);
  }
  for (int t = 0; t < L; ++t) {
    for (int i = 0; i < 8; ++i) out[t][i] = st[t][i];
    if (sum && sum[t]) *sum[t] = so[t];
  }
}
// one step of a path: the running digest on the left or on the right of its sibling
template <int L>
void compress_logged(const Fp (*cur)[8], const Fp (*sib)[8], const bool* cur_right, Fp (*out)[8], const P2Consts* kc,
                     LeafCheckLog* const* log, const uint32_t* kind, const uint32_t* tag, const uint32_t* key, uint32_t mask) {
  Fp st[L][16];
  for (int t = 0; t < L; ++t) {
    for (int i = 0; i < 8; ++i) {
      st[t][i] = cur_right[t] ? sib[t][i] : cur[t][i];
      st[t][8 + i] = cur_right[t] ? cur[t][i] : sib[t][i];
    }
    if (log[t]) log_p2_row(log[t], kind[t], tag[t], key[t], mask, st[t]);
  }
  permute_lanes<L>(st, kc);
  for (int t = 0; t < L; ++t)
    for (int i = 0; i < 8; ++i) out[t][i] = st[t][i];
}
// the last logged row ends its run: its digest is compared with root `rid`, its position goes to the query chip
void log_run_end(LeafCheckLog* log, uint32_t rid) {
  uint32_t* r = log->p2_rows.data() + log->p2_rows.size() - kP2RecWords;
  r[0] |= kP2FlagRe;
  r[kP2RecRid] = rid;
}

// Recomputes the root of one mixed-height opening per lane.  rows[t][c]: opened row of chip c (width[c] words).  With a log: the
// opening as a run of the Poseidon2 chip (air_machine.hpp), tagged `tag`, its root named `rid`; hsum[t][lh]: the Horner sum (in
// alpha) of the opened rows of the chips of height 2^lh, which the query chip turns into reduced openings.  False if a lane's
// root differs.
template <int L>
bool mmcs_verify(const RoundShape& sh, const int* logh, const std::vector<std::vector<Fp>>* const* rows, const size_t* cs,
                 const size_t* m_max, const uint32_t* const* path_canon, const Fp root[8], const P2Consts* kc, LeafCheckLog* const* log,
                 const uint32_t* tag, uint32_t rid, const AlphaPows* alpha, Fp4* const* hsum) {
  const int logn = sh.lm + 1;
  const size_t hm = (size_t)1 << sh.lm;
  size_t pos[L];
  for (int t = 0; t < L; ++t) pos[t] = cs[t] * hm + bitrev32((uint32_t)(m_max[t] & (hm - 1)), sh.lm);
  std::vector<Fp> cat[L];
  const Fp* catp[L];
  auto group_row = [&](int group_logn) {  // the rows of the chips of that height, one after the other (the same widths in every lane)
    for (int t = 0; t < L; ++t) {
      cat[t].clear();
      for (int c = 0; c < kNumChips; ++c)
        if (sh.width[c] && logh[c] + 1 == group_logn) cat[t].insert(cat[t].end(), (*rows[t])[c].begin(), (*rows[t])[c].end());
      catp[t] = cat[t].data();
    }
    return !cat[0].empty();
  };
  // the hashes of the injected rows come first, each labelled with the key and mask its injection row will hold
  Fp inj[32][L][8];
  bool has_inj[32];
  {
    uint32_t key[L], mask = 0;
    for (int t = 0; t < L; ++t) key[t] = 1;
    for (int l = 0; l < logn; ++l) {
      for (int t = 0; t < L; ++t) key[t] = 2 * key[t] + (uint32_t)((pos[t] >> l) & 1);
      mask = 2 * mask;
      has_inj[l] = group_row(logn - l - 1);
      if (has_inj[l]) {
        Fp4* sums[L];
        for (int t = 0; t < L; ++t) sums[t] = hsum && hsum[t] ? &hsum[t][logn - l - 2] : nullptr;
        sponge_logged<L>(catp, cat[0].size(), inj[l], kc, log, tag, key, ++mask, false, true, false, alpha, sums);
      }
    }
  }
  Fp cur[L][8];
  if (!group_row(logn)) return false;
  {
    uint32_t key1[L];
    Fp4* sums[L];
    for (int t = 0; t < L; ++t) { key1[t] = 1; sums[t] = hsum && hsum[t] ? &hsum[t][logn - 1] : nullptr; }
    sponge_logged<L>(catp, cat[0].size(), cur, kc, log, tag, key1, 0, true, false, false, alpha, sums);
  }
  uint32_t key[L], mask = 0;
  for (int t = 0; t < L; ++t) key[t] = 1;
  for (int l = 0; l < logn; ++l) {
    Fp sib[L][8], nxt[L][8];
    bool right[L];
    uint32_t kind[L];
    for (int t = 0; t < L; ++t) {
      for (int i = 0; i < 8; ++i) sib[t][i] = Fp::from_canonical(path_canon[t][8 * l + i]);
      right[t] = (pos[t] >> l) & 1;
      key[t] = 2 * key[t] + (right[t] ? 1u : 0u);
      kind[t] = right[t] ? P2K_PR : P2K_PL;
    }
    mask = 2 * mask;
    compress_logged<L>(cur, sib, right, nxt, kc, log, kind, tag, key, mask);
    if (has_inj[l]) {
      bool left[L];
      uint32_t kj[L];
      for (int t = 0; t < L; ++t) { left[t] = false; kj[t] = P2K_J; }
      compress_logged<L>(nxt, inj[l], left, cur, kc, log, kj, tag, key, ++mask);
    } else {
      for (int t = 0; t < L; ++t)
        for (int i = 0; i < 8; ++i) cur[t][i] = nxt[t][i];
    }
  }
  for (int t = 0; t < L; ++t) {
    for (int i = 0; i < 8; ++i)
      if (cur[t][i] != root[i]) return false;
    if (log[t]) log_run_end(log[t], rid);
  }
  return true;
}

// constraint evaluation at zeta: values are extension elements
struct ZetaCtx {
  using F = Fp4;
  const Fp4* loc;  // main columns at zeta
  const Fp4* nxt;  // main columns at zeta * w
  const Fp4* prp;  // preprocessed columns at zeta
  F prep(int col) const { return prp[col]; }
  const P2Consts* p2() const { return &host_p2_consts(); }
  Fp4 first, trans, last, pub_[kNumCpuPub];
  const Fp4* ap;
  int k_ = 0;
  Fp4 acc = Fp4::zero();
  F local(int col) const { return loc[col]; }
  F next(int col) const { return nxt[col]; }
  F is_first() const { return first; }
  F is_trans() const { return trans; }
  F is_last() const { return last; }
  F pub(int which) const { return pub_[which]; }
  F one() const { return Fp4::one(); }
  F k(uint32_t monty) const { return Fp4::from_base(Fp::raw(monty)); }
  void emit(F v) { acc += ap[k_++] * v; }
  void emit_at(int idx, F v) { acc += ap[idx] * v; }  // fixed index spaces (keccak, CPU)
  void set_count(int n) { k_ = n; }
  Fp4 stash_[32];
  void stash(int i, F v) { stash_[i] = v; }
  F stashed(int i) const { return stash_[i]; }
};

Fp4 lf_eval(const LinForm& f, const Fp4* row) {
  Fp4 v = Fp4::from_base(Fp::raw(f.c0));
  for (int i = 0; i < f.n; ++i) v += row[f.col[i]] * Fp::raw(f.coef[i]);
  return v;
}
Fp4 fingerprint(const Interaction& it, const Fp4* row, const Fp4& gamma, const Fp4* bpow) {
  Fp4 f = gamma + Fp4::from_base(Fp::from_canonical((uint32_t)it.bus));
  for (int j = 0; j < it.n_el; ++j) f += bpow[j + 1] * lf_eval(it.el[j], row);
  return f;
}

void vk_digest_of(const uint32_t root_canon[8], uint32_t entry, uint32_t pad_pc, int log_prog, int log_image, int mode,
                  uint32_t out[8]) {
  Fp v[18];
  for (int i = 0; i < 8; ++i) v[i] = Fp::from_canonical(root_canon[i]);
  const uint32_t rest[10] = {entry & 0xffff, entry >> 16, (uint32_t)log_prog, (uint32_t)log_image, (uint32_t)mode,
                             kMachineVersion, (uint32_t)kCpuWidth, (uint32_t)kNumChips, pad_pc & 0xffff, pad_pc >> 16};
  for (int i = 0; i < 10; ++i) v[8 + i] = Fp::from_canonical(rest[i]);
  Fp d[8];
  hash_elems(v, 18, d, &host_p2_consts());
  for (int i = 0; i < 8; ++i) out[i] = d[i].to_canonical();
}

}  // namespace

namespace {
struct AggNode {
  uint32_t key;
  int have;  // 1 supplied, 2 computed, 0 pending
  Fp d[8];
};
// supplied keys and their ancestors, one entry per key, ascending; false if malformed
bool agg_collect(const uint32_t* keys, size_t n, std::vector<AggNode>* out) {
  std::vector<AggNode>& v = *out;
  v.clear();
  v.reserve(2 * n + 64);
  for (size_t j = 0; j < n; ++j) {
    const uint32_t key = keys ? keys[j] : (uint32_t)(n + j);
    if (key < 2 || key >= (1u << 30)) return false;
    AggNode a{};
    a.key = key; a.have = 1;
    v.push_back(a);
  }
  std::sort(v.begin(), v.end(), [](const AggNode& x, const AggNode& y) { return x.key < y.key; });
  for (size_t j = 1; j < v.size(); ++j)
    if (v[j].key == v[j - 1].key) return false;
  std::vector<uint32_t> anc;
  for (size_t j = 0; j < n; ++j)
    for (uint32_t k = v[j].key >> 1; k >= 1; k >>= 1) {
      anc.push_back(k);
      if (k == 1) break;
    }
  std::sort(anc.begin(), anc.end());
  anc.erase(std::unique(anc.begin(), anc.end()), anc.end());
  for (uint32_t k : anc) {
    if (std::binary_search(v.begin(), v.begin() + n, AggNode{k, 0, {}}, [](const AggNode& x, const AggNode& y) { return x.key < y.key; }))
      return false;  // a supplied node may not be an ancestor of another
    AggNode a{};
    a.key = k; a.have = 0;
    v.push_back(a);
  }
  std::sort(v.begin(), v.end(), [](const AggNode& x, const AggNode& y) { return x.key < y.key; });
  return true;
}
AggNode* agg_find(std::vector<AggNode>& v, uint32_t key) {
  auto it = std::lower_bound(v.begin(), v.end(), key, [](const AggNode& x, uint32_t k) { return x.key < k; });
  return it != v.end() && it->key == key ? &*it : nullptr;
}
}  // namespace

size_t machine_agg_row_count(const uint32_t* keys, size_t n) {
  if (n == 0) return 0;
  std::vector<AggNode> v;
  if (!agg_collect(keys, n, &v)) return SIZE_MAX;
  size_t rows = 0;
  for (AggNode& a : v) {
    if (a.have) continue;
    const AggNode *l = agg_find(v, 2 * a.key), *r = agg_find(v, 2 * a.key + 1);
    if (!l || !r) return SIZE_MAX;
    ++rows;
  }
  return rows;
}

bool machine_nodes_public(const uint32_t* keys, const uint32_t* digests, size_t n, uint32_t root[8], uint32_t list_digest[8],
                          std::vector<uint32_t>* rows) {
  memset(root, 0, 32);
  memset(list_digest, 0, 32);
  if (rows) rows->clear();
  if (n == 0) return true;
  if (!digests) return false;
  for (size_t i = 0; i < 8 * n; ++i)
    if (digests[i] >= kP) return false;
  std::vector<AggNode> v;
  if (!agg_collect(keys, n, &v)) return false;
  for (size_t j = 0; j < n; ++j) {
    AggNode* a = agg_find(v, keys ? keys[j] : (uint32_t)(n + j));
    for (int i = 0; i < 8; ++i) a->d[i] = Fp::from_canonical(digests[8 * j + i]);
  }
  const P2Consts* kc = &host_p2_consts();
  size_t n_rows = 0;
  for (size_t j = v.size(); j-- > 0;) {  // descending keys: children before parents
    if (v[j].have) continue;
    AggNode *l = agg_find(v, 2 * v[j].key), *r = agg_find(v, 2 * v[j].key + 1);
    if (!l || !r || !l->have || !r->have) return false;
    Fp in[16];
    for (int i = 0; i < 8; ++i) { in[i] = l->d[i]; in[8 + i] = r->d[i]; }
    compress(in, in + 8, v[j].d, kc);
    v[j].have = 2;
    ++n_rows;
  }
  if (n_rows == 0 || v[0].key != 1) return false;
  for (int i = 0; i < 8; ++i) root[i] = v[0].d[i].to_canonical();
  std::vector<Fp> flat(9 * n);
  for (size_t j = 0; j < n; ++j) {
    flat[9 * j] = Fp::from_canonical(keys ? keys[j] : (uint32_t)(n + j));
    for (int i = 0; i < 8; ++i) flat[9 * j + 1 + i] = Fp::from_canonical(digests[8 * j + i]);
  }
  Fp dg[8];
  hash_elems(flat.data(), 9 * n, dg, kc);
  for (int i = 0; i < 8; ++i) list_digest[i] = dg[i].to_canonical();
  if (rows) {
    rows->reserve(kP2RecWords * n_rows);
    for (AggNode& a : v) {  // ascending keys: the root first
      if (a.have != 2) continue;
      rows->push_back(P2K_NODE); rows->push_back(0u); rows->push_back(a.key); rows->push_back(0u);  // kind, tag, key, mask
      const AggNode *l = agg_find(v, 2 * a.key), *r = agg_find(v, 2 * a.key + 1);
      for (int i = 0; i < 8; ++i) rows->push_back(l->d[i].to_canonical());
      for (int i = 0; i < 8; ++i) rows->push_back(r->d[i].to_canonical());
      for (int i = 20; i < (int)kP2RecWords; ++i) rows->push_back(0u);  // (root id, Horner sum, alpha_f: a node row has none)
    }
  }
  return true;
}

void machine_prep_traces(const MachineProgram& prog, std::vector<uint32_t>* image_prep, std::vector<uint32_t>* program_prep,
                         std::vector<uint32_t>* table_prep) {
  const size_t hi = (size_t)1 << prog.log_image, hp = (size_t)1 << prog.log_prog, ht = (size_t)1 << kTableLogH;
  image_prep->assign((size_t)kImagePrepWidth * hi, 0);
  program_prep->assign((size_t)kProgramPrepWidth * hp, 0);
  table_prep->assign((size_t)kTablePrepWidth * ht, 0);
  for (size_t r = 0; r < ht; ++r) {
    (*table_prep)[(size_t)TB_P_X * ht + r] = (uint32_t)(r & 255);
    (*table_prep)[(size_t)TB_P_Y * ht + r] = (uint32_t)(r >> 8);
    (*table_prep)[(size_t)TB_P_NA * ht + r] = (r & 3) != 0;
    (*table_prep)[(size_t)TB_P_NT * ht + r] = r == 0 || r > kAddrHiMax;
    (*table_prep)[(size_t)TB_P_XOR * ht + r] = (uint32_t)((r & 255) ^ (r >> 8));
    (*table_prep)[(size_t)TB_P_AND * ht + r] = (uint32_t)((r & 255) & (r >> 8));
  }
  for (size_t r = 0; r < prog.image.size(); ++r) {
    (*image_prep)[(size_t)IMG_P_ADDR * hi + r] = prog.image[r].addr;
    (*image_prep)[(size_t)IMG_P_LO * hi + r] = prog.image[r].val & 0xffff;
    (*image_prep)[(size_t)IMG_P_HI * hi + r] = prog.image[r].val >> 16;
    (*image_prep)[(size_t)IMG_P_REAL * hi + r] = 1;
  }
  for (size_t r = 0; r < prog.rows.size(); ++r) {
    const ProgramRow& p = prog.rows[r];
    uint32_t* q = program_prep->data() + r;
    q[(size_t)PR_PC * hp] = p.pc; q[(size_t)PR_CLS * hp] = (uint32_t)class_of(p.op); q[(size_t)PR_CODE * hp] = code_of(p.op);
    q[(size_t)PR_UC * hp] = ucmp_of(p.op) ? 1u : 0u;
    const bool ecall = p.op == ECALL;  // its CPU row moves t0 only: a0 and a1 are the ecall chip's reads
    q[(size_t)PR_WR * hp] = p.wr; q[(size_t)PR_USE2 * hp] = ecall ? 0 : p.use2;
    q[(size_t)PR_RD * hp] = p.rd; q[(size_t)PR_RS1 * hp] = p.rs1; q[(size_t)PR_RS2 * hp] = ecall ? 0 : p.rs2;
    q[(size_t)PR_IMM_LO * hp] = p.imm & 0xffff; q[(size_t)PR_IMM_HI * hp] = p.imm >> 16;
    q[(size_t)PR_TGT_LO * hp] = p.tgt & 0xffff; q[(size_t)PR_TGT_HI * hp] = p.tgt >> 16;
  }
}

void machine_host_setup(const MachineProgram& prog, MachineVk* vk) {
  const P2Consts* kc = &host_p2_consts();
  constexpr int kPrepMats = 3;
  std::vector<uint32_t> tr[kPrepMats];
  machine_prep_traces(prog, &tr[0], &tr[1], &tr[2]);
  const int logs[kPrepMats] = {prog.log_image, prog.log_prog, kTableLogH},
            widths[kPrepMats] = {kImagePrepWidth, kProgramPrepWidth, kTablePrepWidth};
  std::vector<std::vector<Fp>> lde[kPrepMats];  // [matrix][col] -> [2][H]
  for (int mtx = 0; mtx < kPrepMats; ++mtx) {
    const size_t h = (size_t)1 << logs[mtx];
    lde[mtx].resize(widths[mtx]);
    for (int c = 0; c < widths[mtx]; ++c) {
      std::vector<Fp> col(h);
      for (size_t r = 0; r < h; ++r) col[r] = Fp::from_canonical(tr[mtx][(size_t)c * h + r]);
      host_lde(col, logs[mtx], &lde[mtx][c]);
    }
  }
  // mixed-height tree over (image, program, table) in chip order
  const int lm = std::max(std::max(logs[0], logs[1]), logs[2]), logn = lm + 1;
  auto group_hash = [&](int group_logn, size_t pos, Fp out[8]) -> bool {
    std::vector<Fp> cat;
    for (int mtx = 0; mtx < kPrepMats; ++mtx) {
      if (logs[mtx] + 1 != group_logn) continue;
      const size_t h = (size_t)1 << logs[mtx], cs = pos >> logs[mtx], m = bitrev32((uint32_t)(pos & (h - 1)), logs[mtx]);
      for (int c = 0; c < widths[mtx]; ++c) cat.push_back(lde[mtx][c][cs * h + m]);
    }
    if (cat.empty()) return false;
    hash_elems(cat.data(), cat.size(), out, kc);
    return true;
  };
  std::vector<Fp> level((size_t)8 << logn), nxt;
  for (size_t p = 0; p < ((size_t)1 << logn); ++p) group_hash(logn, p, &level[8 * p]);
  for (int l = 1; l <= logn; ++l) {
    const size_t cnt = (size_t)1 << (logn - l);
    nxt.assign(8 * cnt, Fp::zero());
    for (size_t p = 0; p < cnt; ++p) {
      Fp d[8], g[8];
      compress(&level[16 * p], &level[16 * p + 8], d, kc);
      if (group_hash(logn - l, p, g)) compress(d, g, &nxt[8 * p], kc);
      else for (int i = 0; i < 8; ++i) nxt[8 * p + i] = d[i];
    }
    level.swap(nxt);
  }
  for (int i = 0; i < 8; ++i) vk->prep_root[i] = level[i].to_canonical();
  vk->entry = prog.entry;
  vk->pad_pc = prog.pad_pc();
  vk->log_prog = prog.log_prog;
  vk->log_image = prog.log_image;
  vk->keccak_mode = prog.keccak_mode;
  vk_digest_of(vk->prep_root, vk->entry, vk->pad_pc, vk->log_prog, vk->log_image, vk->keccak_mode, vk->digest);
}

// Coefficients of the reduced openings (format v16; DESIGN.md "Reduced openings").  The input
// of height 2^lh at the LDE point x is  sum_r delta^r (H_r(x) - H_r(zeta)) / (x - zeta)  +  sum_{r = main, perm}
// delta^(3 + r) (H_r(x) - H_r(zeta w)) / (x - zeta w),  H_r = Horner's rule in alpha_f over the segment (r, lh): the opened
// rows of the chips of that height in chip order, zero-filled to a multiple of eight words - the words the opening's sponge
// absorbs, in its order.  Word i of a segment of n words: delta^r alpha_f^(8 ceil(n / 8) - 1 - i).
void machine_reduce_exponents(const int* logh, std::vector<uint32_t>* desc) {
  int wr[kNumChips][4], seg_len[32][4], pos[32][4];
  memset(seg_len, 0, sizeof seg_len);
  memset(pos, 0, sizeof pos);
  size_t n_open = 0;
  for (int c = 0; c < kNumChips; ++c) {
    const ChipDef& d = chip_def(c);
    wr[c][0] = d.prep_w; wr[c][1] = d.main_w; wr[c][2] = d.perm_width(); wr[c][3] = quot_width(logh, c);
    for (int r = 0; r < 4; ++r) seg_len[logh[c]][r] += wr[c][r];
    n_open += (size_t)wr[c][0] + 2 * (size_t)wr[c][1] + 2 * (size_t)wr[c][2] + (size_t)wr[c][3];
  }
  desc->assign(n_open, 0);
  size_t off = 0;
  for (int c = 0; c < kNumChips; ++c) {
    const int lh = logh[c];
    const size_t n1 = (size_t)wr[c][0] + wr[c][1] + wr[c][2] + wr[c][3];
    size_t i = 0, j = 0;
    for (int r = 0; r < 4; ++r) {
      const int lpad = (seg_len[lh][r] + 7) / 8 * 8;
      for (int col = 0; col < wr[c][r]; ++col, ++i) {
        const uint32_t e = (uint32_t)(lpad - 1 - (pos[lh][r] + col));
        (*desc)[off + i] = e | ((uint32_t)r << 16);
        if (r == 1 || r == 2) { (*desc)[off + n1 + j] = e | ((uint32_t)(3 + r) << 16); ++j; }
      }
      pos[lh][r] += wr[c][r];
    }
    off += n1 + (size_t)wr[c][1] + wr[c][2];
  }
}
void machine_reduce_coefs(const int* logh, const Fp4& af, const Fp4& delta, Fp4* out) {
  std::vector<uint32_t> desc;
  machine_reduce_exponents(logh, &desc);
  uint32_t emax = 0;
  for (uint32_t d : desc) emax = std::max(emax, d & 0xffffu);
  std::vector<Fp4> ap(emax + 1);
  ap[0] = Fp4::one();
  for (uint32_t i = 1; i <= emax; ++i) ap[i] = ap[i - 1] * af;
  Fp4 dp[6];
  dp[0] = Fp4::one();
  for (int i = 1; i < 6; ++i) dp[i] = dp[i - 1] * delta;
  for (size_t t = 0; t < desc.size(); ++t) out[t] = dp[desc[t] >> 16] * ap[desc[t] & 0xffffu];
}

size_t machine_proof_body_words(const int* logh, uint32_t num_queries) {
  int lm = 0, lm_prep = 0;
  size_t opened = 0, rw[4] = {0, 0, 0, 0};
  for (int c = 0; c < kNumChips; ++c) {
    const ChipDef& d = chip_def(c);
    const size_t e = (size_t)d.perm_width();
    lm = std::max(lm, logh[c]);
    if (d.prep_w) lm_prep = std::max(lm_prep, logh[c]);
    const size_t q = (size_t)quot_width(logh, c);  // one quotient per height: its first chip's
    opened += (size_t)d.prep_w + 2 * (size_t)d.main_w + 2 * e + q;
    rw[0] += (size_t)d.prep_w; rw[1] += (size_t)d.main_w; rw[2] += e; rw[3] += q;
  }
  size_t words = 8 + 8 + 4 * (size_t)kNumChips + 8 + 4 * opened + 8 * (size_t)lm + 4 + 1;
  size_t perq = rw[0] + 8 * ((size_t)lm_prep + 1);
  for (int r = 1; r < 4; ++r) perq += rw[r] + 8 * ((size_t)lm + 1);
  for (int k = 0; k < lm; ++k) perq += 8 + 8 * (size_t)(lm - k);
  return words + perq * num_queries;
}

bool parse_machine_header(const uint8_t* bytes, size_t len, MachineHeader* h, std::string* err) {
  if (len < (size_t)kHeaderWords * 4 || (len & 3)) { *err = "proof too short"; return false; }
  const uint32_t* w = reinterpret_cast<const uint32_t*>(bytes);
  if (w[0] != kProofMagic) { *err = "bad magic"; return false; }
  if (w[1] != kMachineVersion) { *err = "unsupported proof version"; return false; }
  for (int c = 0; c < kNumChips; ++c) {
    h->logh[c] = (int)w[2 + c];
    if (w[2 + c] < 5 || w[2 + c] > 21) { *err = "chip height out of range"; return false; }
  }
  {
    // at most 2^21 CPU rows, as the time limbs of the memory argument assume (times stay below 2^24)
    size_t rows = 0;
    for (int i = 0; i < kNumCpuInst; ++i) rows += (size_t)1 << h->logh[cpu_chip(i)];
    if (rows > ((size_t)1 << 21)) { *err = "CPU instance heights out of range"; return false; }
  }
  h->exit_code = w[2 + kNumChips];
  h->pv_len = w[3 + kNumChips];
  memcpy(h->pv_digest, w + 4 + kNumChips, 32);
  memcpy(h->deferred_digest, w + 12 + kNumChips, 32);
  memcpy(h->vk_digest, w + 20 + kNumChips, 32);
  constexpr int kHo = kNumCpuInst - 1;
  for (int i = 0; i < kHo; ++i) h->handover_pc[i] = w[28 + kNumChips + i];
  h->agg_n = w[28 + kNumChips + kHo];
  memcpy(h->agg_root, w + 29 + kNumChips + kHo, 32);
  memcpy(h->agg_digest, w + 37 + kNumChips + kHo, 32);
  if (h->agg_n == 1 || h->agg_n > (1u << 20)) { *err = "aggregation payload of an impossible size"; return false; }
  for (int i = 0; i < 8; ++i)
    if (h->agg_root[i] >= kP || h->agg_digest[i] >= kP) { *err = "non-canonical aggregation digest"; return false; }
  h->pub_n = w[45 + kNumChips + kHo];
  memcpy(h->pub_digest, w + 46 + kNumChips + kHo, 32);
  if (h->pub_n > (1u << 22)) { *err = "public tuple list of an impossible size"; return false; }
  for (int i = 0; i < 8; ++i)
    if (h->pub_digest[i] >= kP) { *err = "non-canonical public-tuple digest"; return false; }
  if (h->pv_len > (1u << 24)) { *err = "public values too long"; return false; }
  h->pv_offset = (size_t)kHeaderWords * 4;
  h->body_offset = ((size_t)kHeaderWords + (h->pv_len + 3) / 4) * 4;
  if (len < h->body_offset) { *err = "proof truncated in header"; return false; }
  for (size_t i = h->pv_offset + h->pv_len; i < h->body_offset; ++i)
    if (bytes[i] != 0) { *err = "non-zero padding after the public values"; return false; }
  return true;
}

void machine_pub_digest(const uint32_t* pub_tuples, size_t n_pub, uint32_t digest[8]) {
  memset(digest, 0, 32);
  if (n_pub == 0) return;
  std::vector<Fp> flat(n_pub * kPubTupleWords);
  for (size_t i = 0; i < flat.size(); ++i) flat[i] = Fp::from_canonical(pub_tuples[i] % kP);
  Fp dg[8];
  hash_elems(flat.data(), flat.size(), dg, &host_p2_consts());
  for (int i = 0; i < 8; ++i) digest[i] = dg[i].to_canonical();
}

int verify_machine_proof(const uint8_t* bytes, size_t len, const MachineVk& vk, uint32_t num_queries, uint32_t pow_bits,
                         std::string* err, const uint32_t* agg_leaves, size_t n_agg, const uint32_t* agg_keys,
                         const uint32_t* pub_tuples, size_t n_pub, LeafCheckLog* log, bool stub, unsigned max_threads,
                         ZetaSelfTest* zeta_selftest) {
  MachineHeader hd;
  if (!parse_machine_header(bytes, len, &hd, err)) return 7;
  // public bus tuples: the caller names the statement the proof's buses are claimed to close with; the transcript holds its digest
  if (hd.pub_n != n_pub) {
    *err = n_pub ? "the proof does not carry this many public bus tuples" : "the proof carries public bus tuples (a leaf-proof check): verify it with them";
    return 8;
  }
  if (n_pub) {
    for (size_t i = 0; i < n_pub; ++i) {
      const uint32_t* t = pub_tuples + kPubTupleWords * i;
      if (t[1] > 1 || t[3] > (uint32_t)(kPubTupleWords - 4)) { *err = "malformed public bus tuple"; return 7; }
      // only the buses of a leaf-proof check have a public end: what the verifier sends, and the proof-of-work word it takes
      const bool sent = t[0] == BUS_TBLK || t[0] == BUS_TSQ || t[0] == BUS_ROOT || t[0] == BUS_LEAFK || t[0] == BUS_BCONST;
      if (t[1] ? !sent : t[0] != BUS_POW) { *err = "public bus tuple on a bus without a public end"; return 7; }
      for (int j = 0; j < kPubTupleWords; ++j)
        if (t[j] >= kP) { *err = "non-canonical word in a public bus tuple"; return 7; }
    }
    uint32_t dg[8];
    machine_pub_digest(pub_tuples, n_pub, dg);
    if (memcmp(dg, hd.pub_digest, 32) != 0) { *err = "the proof was made for another list of public bus tuples"; return 8; }
  }
  if (log) { log->p2_rows.clear(); log->tr_rows.clear(); log->qr_rows.clear(); log->pub_tuples.clear(); }
  if (log && (num_queries > kLeafMaxQueries || log->leaf_index >= 4096)) { *err = "leaf check: too many queries or leaves for the tag space"; return 7; }
  // the aggregation payload: the caller names the leaves the proof's root is claimed for; the transcript holds their digest
  if (hd.agg_n != n_agg) {
    *err = n_agg ? "the proof does not aggregate this many leaves" : "the proof carries an aggregation payload: verify it with its leaves";
    return 8;
  }
  if (n_agg) {
    uint32_t r[8], dg[8];
    if (!machine_nodes_public(agg_keys, agg_leaves, n_agg, r, dg, nullptr)) { *err = "malformed aggregation leaves"; return 7; }
    if (memcmp(dg, hd.agg_digest, 32) != 0) { *err = "the proof aggregates another list of leaves"; return 8; }
  }
  const int* logh = hd.logh;
  // (a stub is a proof cut off in front of its query phase: everything the transcript absorbs, and the opened values)
  const size_t body_words = stub ? machine_proof_body_words(logh, 0) : machine_proof_body_words(logh, num_queries);
  if (len != hd.body_offset + body_words * 4) { *err = stub ? "proof stub length mismatch" : "proof length mismatch"; return 7; }
  if (memcmp(hd.vk_digest, vk.digest, 32) != 0) { *err = "verifying key mismatch"; return 8; }
  if (logh[kImage] != vk.log_image || logh[kProgram] != vk.log_prog || logh[kTable] != kTableLogH) {
    *err = "preprocessed table heights differ from the key";
    return 8;
  }
  {
    uint8_t dg[32];
    sha256(bytes + hd.pv_offset, hd.pv_len, dg);
    if (memcmp(dg, hd.pv_digest, 32) != 0) { *err = "public-values digest mismatch"; return 8; }
  }
  if (hd.exit_code != 0) { *err = "guest exit code is not zero"; return 8; }
  const uint32_t* body = reinterpret_cast<const uint32_t*>(bytes + hd.body_offset);
  for (size_t i = 0; i < body_words; ++i)
    if (body[i] >= kP) { *err = "non-canonical field element"; return 7; }

  const P2Consts* kc = &host_p2_consts();
  int lm = 0;  // the tallest chip: every tree and the FRI start from its height
  for (int c = 0; c < kNumChips; ++c) lm = std::max(lm, logh[c]);
  // shapes
  RoundShape shape[4];
  size_t open_off[kNumChips], n_open = 0;
  for (int r = 0; r < 4; ++r) shape[r].lm = 0;
  for (int c = 0; c < kNumChips; ++c) {
    const ChipDef& d = chip_def(c);
    const int w[4] = {d.prep_w, d.main_w, d.perm_width(), quot_width(logh, c)};
    for (int r = 0; r < 4; ++r) {
      shape[r].width[c] = w[r];
      if (w[r]) shape[r].lm = std::max(shape[r].lm, logh[c]);
    }
    open_off[c] = n_open;
    n_open += (size_t)d.prep_w + 2 * (size_t)d.main_w + 2 * (size_t)d.perm_width() + (size_t)quot_width(logh, c);
  }
  const uint32_t* p_root_main = body;
  const uint32_t* p_root_perm = body + 8;
  const uint32_t* p_cum = body + 16;
  const uint32_t* p_root_quot = p_cum + 4 * kNumChips;
  const uint32_t* p_opened = p_root_quot + 8;
  const uint32_t* p_fri_roots = p_opened + 4 * n_open;
  const uint32_t* p_final = p_fri_roots + 8 * (size_t)lm;
  const uint32_t* p_witness = p_final + 4;
  const uint32_t* p_queries = p_witness + 1;

  // ---- transcript ----
  HostChallenger ch(kc);
  std::vector<uint32_t> duplexes;  // with a log: every duplex of this transcript (17 words each), for the transcript chip
  if (log) ch.record = &duplexes;
  for (int i = 0; i < 8; ++i) ch.observe_canon(hd.vk_digest[i]);
  for (int c = 0; c < kNumChips; ++c) ch.observe_canon((uint32_t)logh[c]);
  ch.observe_canon(hd.exit_code & 0xffff);
  ch.observe_canon(hd.exit_code >> 16);
  for (int i = 0; i < 8; ++i) { ch.observe_canon(hd.pv_digest[i] & 0xffff); ch.observe_canon(hd.pv_digest[i] >> 16); }
  for (int i = 0; i < 8; ++i) { ch.observe_canon(hd.deferred_digest[i] & 0xffff); ch.observe_canon(hd.deferred_digest[i] >> 16); }
  for (int i = 0; i < kNumCpuInst - 1; ++i) {
    ch.observe_canon(hd.handover_pc[i] & 0xffff);
    ch.observe_canon(hd.handover_pc[i] >> 16);
  }
  ch.observe_canon(hd.agg_n);
  for (int i = 0; i < 8; ++i) ch.observe_canon(hd.agg_root[i]);
  for (int i = 0; i < 8; ++i) ch.observe_canon(hd.agg_digest[i]);
  ch.observe_canon(hd.pub_n);
  for (int i = 0; i < 8; ++i) ch.observe_canon(hd.pub_digest[i]);
  ch.pad();  // (v16: a commitment root is a block of its own)
  Fp root[4][8];
  for (int i = 0; i < 8; ++i) root[0][i] = Fp::from_canonical(vk.prep_root[i]);
  for (int i = 0; i < 8; ++i) { root[1][i] = Fp::from_canonical(p_root_main[i]); ch.observe(root[1][i]); }
  ch.pad();  // (v16: a phase of the transcript ends on a block boundary)
  const Fp4 gamma = ch.sample_ext(), beta = ch.sample_ext();
  for (int i = 0; i < 8; ++i) { root[2][i] = Fp::from_canonical(p_root_perm[i]); ch.observe(root[2][i]); }
  Fp4 cum[kNumChips];
  for (int c = 0; c < kNumChips; ++c) {
    cum[c] = read_fp4(p_cum + 4 * c);
    for (int i = 0; i < 4; ++i) ch.observe(cum[c].c[i]);
  }
  ch.pad();
  const Fp4 alpha = ch.sample_ext();
  for (int i = 0; i < 8; ++i) { root[3][i] = Fp::from_canonical(p_root_quot[i]); ch.observe(root[3][i]); }
  const Fp4 zeta = ch.sample_ext();

  // ---- the buses balance: chips + the public COMMIT / COMMIT_DEFERRED / HALT terms ----
  Fp4 bpow[kInterMaxElems + 1];
  bpow[0] = Fp4::one();
  for (int j = 1; j <= kInterMaxElems; ++j) bpow[j] = bpow[j - 1] * beta;
  {
    Fp4 total = Fp4::zero();
    for (int c = 0; c < kNumChips; ++c) total += cum[c];
    auto fc = [](uint32_t v) { return Fp4::from_base(Fp::from_canonical(v)); };
    for (uint32_t kind = 1; kind <= 2; ++kind)
      for (uint32_t i = 0; i < 8; ++i) {
        const uint32_t w = kind == 1 ? hd.pv_digest[i] : hd.deferred_digest[i];
        const Fp4 f = gamma + fc(BUS_PUBC) + bpow[1] * fc(kind) + bpow[2] * fc(i) + bpow[3] * fc(w & 0xffff) + bpow[4] * fc(w >> 16);
        total -= f.inv();
      }
    const Fp4 fh = gamma + fc(BUS_PUBH) + bpow[1] * fc(hd.exit_code & 0xffff) + bpow[2] * fc(hd.exit_code >> 16);
    total -= fh.inv();
    // ... and the digest bus of the aggregation payload: the supplied digests go in at their heap keys (the leaves of a full
    // tree at n .. 2n - 1, or a leaf and the siblings of its path), the root comes out at 1
    // (DIGEST tuples: tag 0, type 2 = heap node, key, mask 0, the digest)
    for (size_t i = 0; i <= n_agg && n_agg; ++i) {
      const uint32_t* d = i < n_agg ? agg_leaves + 8 * i : hd.agg_root;
      Fp4 f = gamma + fc(BUS_DIGEST) + bpow[2] * fc(2u) + bpow[3] * fc(i < n_agg ? (agg_keys ? agg_keys[i] : (uint32_t)(n_agg + i)) : 1u);
      for (int j = 0; j < 8; ++j) f += bpow[5 + j] * fc(d[j]);
      if (i < n_agg) total += f.inv();
      else total -= f.inv();
    }
    // ... and the public bus tuples of a leaf-proof check, each sent or received by the verifier with its multiplicity
    for (size_t i = 0; i < n_pub; ++i) {
      const uint32_t* t = pub_tuples + kPubTupleWords * i;
      Fp4 f = gamma + fc(t[0]);
      for (uint32_t j = 0; j < t[3]; ++j) f += bpow[1 + j] * fc(t[4 + j]);
      const Fp4 term = f.inv() * Fp::from_canonical(t[2]);
      if (t[1]) total += term;
      else total -= term;
    }
    if (total != Fp4::zero()) { *err = "LogUp buses do not balance against the public values and exit code"; return 8; }
  }

  const unsigned n_thr = std::max(1u, std::min({std::thread::hardware_concurrency(), max_threads ? max_threads : 8u, (unsigned)num_queries / 4u}));
  std::vector<Fp4> opened(n_open);
  for (size_t i = 0; i < n_open; ++i) opened[i] = read_fp4(p_opened + 4 * i);
  {
    std::vector<Fp> words(n_open * 4);
    for (size_t t = 0; t < n_open * 4; ++t) words[t] = Fp::from_canonical(p_opened[t]);
    Fp open_root[8];
    list_root(words, ceil_log2((n_open * 4 + 7) / 8), open_root, kc, n_thr);  // (8 000 permutations: a third of a verification on one thread)
    for (int i = 0; i < 8; ++i) ch.observe(open_root[i]);
  }
  const Fp4 af = ch.sample_ext(), delta = ch.sample_ext();

  // ---- constraint identity at zeta: one per height (the chips of a height share a quotient: machine_defs.hpp) ----
  const Fp g = Fp::from_canonical(kGen);
  Fp4 group_acc[kNumChips];  // indexed by the height's first chip
  for (int c = 0; c < kNumChips; ++c) group_acc[c] = Fp4::zero();
  // (zeta_selftest: the same identity as the recorded program of zeta_program.hpp, on this proof's values)
  const ZetaProgram* zprog = zeta_selftest ? &zeta_program() : nullptr;
  std::vector<Fp4> zcells(zprog ? zprog->n_cells : 0, Fp4::zero());
  Fp4 chip_acc[kNumChips];
  for (int c = 0; c < kNumChips; ++c) {
    const ChipDef& d = chip_def(c);
    const int pw = d.prep_w, mw = d.main_w, ew = d.perm_width(), nh = d.helpers(), nb = d.n_constraints, qw = quot_width(logh, c);
    const size_t h = (size_t)1 << logh[c];
    const Fp4* o_prep = opened.data() + open_off[c];
    const Fp4* o_main = o_prep + pw;
    const Fp4* o_perm = o_main + mw;
    const Fp4* o_quot = o_perm + ew;
    const Fp4* o_main_n = o_quot + qw;
    const Fp4* o_perm_n = o_main_n + mw;
    const Fp wh = fp_root_of_unity(logh[c]), wh_inv = wh.inv();
    const Fp4 zeta_h = zeta.pow(h), zh = zeta_h - Fp4::one();
    std::vector<Fp4> apow(d.total_constraints());
    apow[0] = alpha.pow((uint64_t)quot_alpha_offset(logh, c));
    for (size_t k = 1; k < apow.size(); ++k) apow[k] = apow[k - 1] * alpha;
    ZetaCtx zc;
    zc.loc = o_main;
    zc.nxt = o_main_n;
    zc.prp = o_prep;
    zc.first = zh * (zeta - Fp4::one()).inv();
    zc.trans = zeta - Fp4::from_base(wh_inv);
    zc.last = zh * (zeta - Fp4::from_base(wh_inv)).inv();
    // the CPU instances' public scalars: the first starts at the entry point at time 4 and hands over to the second,
    // which starts at the hand-over pc (a header word, absorbed into the transcript) right after the first's last row
    // and ends on the padding instruction (reached through HALT only: the exit code bus would not balance otherwise)
    for (int i = 0; i < kNumCpuPub; ++i) zc.pub_[i] = Fp4::zero();
    zc.pub_[kPubPadPc] = Fp4::from_base(Fp::from_canonical(vk.pad_pc));
    if (is_cpu_chip(c)) {
      const int inst = cpu_instance(c);
      size_t r0 = 0;
      for (int i = 0; i < inst; ++i) r0 += (size_t)1 << logh[cpu_chip(i)];
      zc.pub_[kPubStartPc] = Fp4::from_base(Fp::from_canonical(inst == 0 ? vk.entry : hd.handover_pc[inst - 1] % kP));
      zc.pub_[kPubStartTs] = Fp4::from_base(Fp::from_canonical((uint32_t)(4 * (r0 + 1))));
      if (inst + 1 < kNumCpuInst) {
        zc.pub_[kPubHasSucc] = Fp4::one();
        zc.pub_[kPubEndPc] = Fp4::from_base(Fp::from_canonical(hd.handover_pc[inst] % kP));
      }
    }
    zc.ap = apow.data();
    switch (c) {
      case kCpu:
      case kCpu2: case kCpu3: case kCpu4: case kCpu5: case kCpu6: case kCpu7: case kCpu8: eval_cpu(zc); break;
      case kKeccak:
        for (int task = 0; task < ka::kBusTask; ++task) ka::eval_task(task, zc);
        zc.k_ = ka::kNumConstraints;
        eval_keccak_ts(zc);
        break;
      case kKmem: eval_kmem(zc); break;
      case kMemFinal: eval_memfinal(zc); break;
      case kImage: eval_image(zc); break;
      case kProgram: break;
      case kMul: eval_mul(zc); break;
      case kTable: eval_table(zc); break;
      case kAlu:
      case kAlu2: eval_alu(zc); break;
      case kSub:
      case kSub2: eval_sub(zc); break;
      case kBw:
      case kBw2: eval_bw(zc); break;
      case kP2: eval_p2(zc); break;
      case kEcall: eval_ecall(zc); break;
      case kQr: eval_qr(zc); break;
      case kTr: eval_tr(zc); break;
      case kDiv: eval_div(zc); break;
      case kHint: eval_hint(zc); break;
    }
    if (zc.k_ != nb) { *err = "internal: constraint count"; return 7; }
    // LogUp (machine_defs.hpp "LogUp layout"): row = [prep | main] at zeta
    std::vector<Fp4> row(o_prep, o_prep + pw + mw);
    const int nr = d.n_inter - d.n_merged, npairs = (nr + 1) / 2;
    auto signed_mult = [&](const Interaction& it) {
      const Fp4 m = lf_eval(it.mult, row.data());
      return it.sign < 0 ? -m : m;
    };
    // v fa fb - (ma fb + mb fa) for slot s and a candidate value v
    auto slot_constraint = [&](int s, const Fp4& v) {
      Fp4 ma, fa, mb = Fp4::zero(), fb = Fp4::one();
      if (s < npairs) {
        ma = signed_mult(d.inter[2 * s]);
        fa = fingerprint(d.inter[2 * s], row.data(), gamma, bpow);
        if (2 * s + 1 < nr) {
          mb = signed_mult(d.inter[2 * s + 1]);
          fb = fingerprint(d.inter[2 * s + 1], row.data(), gamma, bpow);
        }
      } else {  // the merged sends: (sum m_k) / (sum m_k f_k + 1 - sum m_k)
        ma = Fp4::zero();
        fa = Fp4::zero();
        for (int k = nr; k < d.n_inter; ++k) {
          const Fp4 m = signed_mult(d.inter[k]);
          ma += m;
          fa += m * fingerprint(d.inter[k], row.data(), gamma, bpow);
        }
        fa += Fp4::one() - ma;
      }
      return v * fa * fb - (fb * ma + fa * mb);
    };
    Fp4 hsum = Fp4::zero();
    for (int j = 0; j < nh; ++j) {
      const Fp4 hj = from_basis(o_perm + 4 * j);
      hsum += hj;
      zc.acc += apow[nb + j] * slot_constraint(j, hj);
    }
    const Fp4 phi = from_basis(o_perm + 4 * nh), phin = from_basis(o_perm_n + 4 * nh);
    const Fp4 cum_step = cum[c] * Fp::from_canonical((uint32_t)(h % kP)).inv();
    zc.acc += apow[nb + nh] * slot_constraint(nh, phin - phi + cum_step - hsum);
    group_acc[quot_leader(logh, c)] += zc.acc;
    chip_acc[c] = zc.acc;
    if (zprog) {  // this chip's inputs of the program
      const ZetaChipCells& cc = zprog->chip[c];
      for (int i = 0; i < pw; ++i) zcells[cc.prep + i] = o_prep[i];
      for (int i = 0; i < mw; ++i) { zcells[cc.main + i] = o_main[i]; zcells[cc.main_next + i] = o_main_n[i]; }
      for (int i = 0; i < ew; ++i) zcells[cc.perm + i] = o_perm[i];
      for (int i = 0; i < 4; ++i) zcells[cc.perm_next_phi + i] = o_perm_n[4 * nh + i];
      for (int i = 0; i < qw; ++i) zcells[cc.quot + i] = o_quot[i];
      zcells[cc.first] = zc.first; zcells[cc.trans] = zc.trans; zcells[cc.last] = zc.last;
      zcells[cc.apow0] = apow[0];
      zcells[cc.cum_step] = cum_step;
      for (int i = 0; i < kNumCpuPub; ++i) zcells[cc.pub[i]] = zc.pub_[i];
    }
  }
  for (int c = 0; c < kNumChips; ++c) {
    if (!quot_width(logh, c)) continue;
    const ChipDef& d = chip_def(c);
    const size_t h = (size_t)1 << logh[c];
    const Fp4* o_quot = opened.data() + open_off[c] + d.prep_w + d.main_w + d.perm_width();
    const Fp4 zeta_h = zeta.pow(h), zh = zeta_h - Fp4::one();
    const Fp4 q0 = from_basis(o_quot), q1 = from_basis(o_quot + 4);
    const Fp sh = g.pow(h), inv_2sh = (sh + sh).inv();
    const Fp4 quot = q0 * (zeta_h + Fp4::from_base(sh)) * inv_2sh - q1 * (zeta_h - Fp4::from_base(sh)) * inv_2sh;
    if (group_acc[c] != quot * zh) {
      *err = std::string("constraint identity fails at zeta for the chips of height 2^") + std::to_string(logh[c]) + " (first: " + d.name + ")";
      return 8;
    }
  }

  if (zprog) {
    // the final combination's weights: kappa_c = delta^(the place of c's height among the heights), and the recomposition
    // coefficients of the height's two quotient chunks for the chip that carries them
    Fp4 native = Fp4::zero(), kap = Fp4::one();
    Fp4 kappa_of[kNumChips];
    for (int c = 0; c < kNumChips; ++c) {
      const ZetaChipCells& cc = zprog->chip[c];
      if (quot_leader(logh, c) == c) {
        const ChipDef& d = chip_def(c);
        const size_t h = (size_t)1 << logh[c];
        const Fp4* o_quot = opened.data() + open_off[c] + d.prep_w + d.main_w + d.perm_width();
        const Fp4 zeta_h = zeta.pow(h), zh = zeta_h - Fp4::one();
        const Fp sh = g.pow(h), inv_2sh = (sh + sh).inv();
        kappa_of[c] = kap;
        zcells[cc.u] = kap * zh * (zeta_h + Fp4::from_base(sh)) * inv_2sh;
        zcells[cc.v] = kap * zh * (zeta_h - Fp4::from_base(sh)) * inv_2sh;
        const Fp4 q0 = from_basis(o_quot), q1 = from_basis(o_quot + 4);
        const Fp4 quot = q0 * (zeta_h + Fp4::from_base(sh)) * inv_2sh - q1 * (zeta_h - Fp4::from_base(sh)) * inv_2sh;
        native += kap * (group_acc[c] - quot * zh);
        kap = kap * delta;
      }
      zcells[cc.kappa] = kappa_of[quot_leader(logh, c)];
    }
    zcells[zprog->alpha] = alpha; zcells[zprog->gamma] = gamma; zcells[zprog->beta] = beta;
    zeta_program_run(*zprog, zcells.data());
    zeta_selftest->n_ops = (uint32_t)zprog->ops.size();
    zeta_selftest->n_cells = zprog->n_cells;
    zeta_selftest->n_inputs = zprog->n_inputs;
    zeta_selftest->n_consts = (uint32_t)zprog->const_cell.size();
    zeta_selftest->max_reads = zprog->max_reads;
    zeta_selftest->inputs_read = zprog->inputs_read;
    // (the arithmetic chip's memory argument on these values: every cell written once, read as often as the program says)
    if (!zeta_program_memory_balances(*zprog, zcells.data(), gamma, beta)) { *err = "internal: the zeta program's memory does not balance"; return 7; }
    for (int c = 0; c < kNumChips && zeta_selftest->mismatch_chip < 0; ++c)
      if (zcells[zprog->chip[c].acc] != chip_acc[c]) zeta_selftest->mismatch_chip = c;
    if (zeta_selftest->mismatch_chip < 0 && (zcells[zprog->result] != native || native != Fp4::zero())) zeta_selftest->mismatch_chip = kNumChips;
    if (zeta_selftest->mismatch_chip >= 0) { *err = "internal: the recorded zeta program disagrees with the native evaluation"; return 7; }
  }
  // ---- FRI transcript ----
  std::vector<Fp4> betas(lm);
  std::vector<std::array<Fp, 8>> fri_roots(lm);
  for (int k = 0; k < lm; ++k) {
    for (int i = 0; i < 8; ++i) { fri_roots[k][i] = Fp::from_canonical(p_fri_roots[8 * k + i]); ch.observe(fri_roots[k][i]); }
    betas[k] = ch.sample_ext();
  }
  const Fp4 final_poly = read_fp4(p_final);
  for (int i = 0; i < 4; ++i) ch.observe(final_poly.c[i]);
  ch.observe_canon(p_witness[0]);
  ch.pad();
  const uint32_t pow_word = ch.sample().to_canonical();
  if ((pow_word & ((1u << pow_bits) - 1)) != 0) { *err = "proof-of-work witness rejected"; return 8; }
  ch.drop_outputs();  // (v16: the query indices start from a fresh squeeze)

  // ---- reduced-opening constants per chip (v16: machine_reduce_coefs) ----
  std::vector<Fp4> afpow(n_open);
  machine_reduce_coefs(logh, af, delta, afpow.data());
  Fp4 b1[kNumChips], b2[kNumChips];
  size_t n1[kNumChips], n2[kNumChips];
  for (int c = 0; c < kNumChips; ++c) {
    const ChipDef& d = chip_def(c);
    n1[c] = (size_t)d.prep_w + d.main_w + d.perm_width() + (size_t)quot_width(logh, c);
    n2[c] = (size_t)d.main_w + d.perm_width();
    b1[c] = b2[c] = Fp4::zero();
    for (size_t i = 0; i < n1[c]; ++i) b1[c] += afpow[open_off[c] + i] * opened[open_off[c] + i];
    for (size_t i = 0; i < n2[c]; ++i) b2[c] += afpow[open_off[c] + n1[c] + i] * opened[open_off[c] + n1[c] + i];
  }
  const Fp inv2 = Fp::from_canonical(2).inv();
  auto height_present = [&](int lh) { for (int c = 0; c < kNumChips; ++c) if (logh[c] == lh) return true; return false; };
  auto height_has_prep = [&](int lh) { for (int c = 0; c < kNumChips; ++c) if (logh[c] == lh && chip_def(c).prep_w) return true; return false; };
  int n_heights = 0;
  for (int lh = 0; lh <= lm; ++lh) n_heights += height_present(lh) ? 1 : 0;

  size_t perq = 0;
  for (int r = 0; r < 4; ++r) {
    for (int c = 0; c < kNumChips; ++c) perq += (size_t)shape[r].width[c];
    perq += 8 * ((size_t)shape[r].lm + 1);
  }
  for (int k = 0; k < lm; ++k) perq += 8 + 8 * (size_t)(lm - k);
  const size_t hmax = (size_t)1 << lm;
  // The queries are independent of one another once their indices are drawn: they are checked on several threads (each with
  // its own log, appended in query order afterwards), the first failure in query order is the one reported.
  std::vector<size_t> indices(num_queries);
  std::vector<uint32_t> index_words(num_queries);
  for (uint32_t qi = 0; qi < num_queries; ++qi) {
    index_words[qi] = ch.sample().to_canonical();
    indices[qi] = index_words[qi] & (((uint32_t)2 << lm) - 1);
  }
  ch.record = nullptr;
  const uint32_t leaf = log ? log->leaf_index : 0u;
  auto canon4 = [](const Fp4& v, uint32_t* out) { for (int i = 0; i < 4; ++i) out[i] = v.c[i].to_canonical(); };

  // ---- with a log: the statement a proof ABOUT this verification closes its buses with, and the transcript chip's rows ----
  if (log) {
    if (shape[0].lm != kTableLogH || lm < kTableLogH || lm > 21) { *err = "leaf check: unsupported shape"; return 7; }
    const size_t n_dup = duplexes.size() / 17;
    // the steps, in the order the transcript went through them
    constexpr size_t kHeadObs = 8 + kNumChips + 2 + 16 + 16 + 2 * (kNumCpuInst - 1) + 1 + 8 + 8 + 1 + 8;  // words observed before the main root
    const size_t s_main = (kHeadObs + 7) / 8, s_perm = s_main + 1, s_quot = s_perm + 1 + (4 * (size_t)kNumChips + 7) / 8, s_open = s_quot + 1,
                 s_fri = s_open + 1, s_final = s_fri + (size_t)lm, s_q = s_final + 1, n_sq = (num_queries + 7) / 8;
    if (n_dup != s_q + n_sq) { *err = "internal: transcript steps"; return 7; }
    for (size_t st = 0; st < n_dup; ++st) {
      const uint32_t* d = duplexes.data() + 17 * st;
      if (d[0] > 1 || (d[0] == 1) != (st < s_q)) { *err = "internal: transcript is not block-aligned"; return 7; }
      uint32_t flags = 0, ridk = 0, qbase = 0, mroot = 0, mfin = 0, mzeta = 0, maf = 0, mbeta = 0;
      if (st == s_main) { flags |= 1; ridk = 1; mroot = num_queries; }
      if (st == s_perm) { flags |= 1; ridk = 2; mroot = num_queries; }
      if (st == s_quot) { flags |= 1 | 2; ridk = 3; mroot = num_queries; mzeta = num_queries * (uint32_t)n_heights; }
      if (st == s_open) { flags |= 4; maf = num_queries * (uint32_t)n_heights; }
      if (st >= s_fri && st < s_final) { flags |= 1 | 8; ridk = 4 + (uint32_t)(st - s_fri); mroot = num_queries; mbeta = num_queries; }
      if (st == s_final) { flags |= 16 | 32; mfin = num_queries; }
      if (st >= s_q) {
        qbase = 8 * (uint32_t)(st - s_q);
        const uint32_t nq = std::min<uint32_t>(8, num_queries - qbase);
        flags |= 64 | (((1u << nq) - 1) << 7);
      }
      std::vector<uint32_t>& v = log->tr_rows;
      v.push_back(flags | (st == 0 ? kTrRecFirst : 0u) | (d[0] ? kTrRecAbs : 0u));
      v.push_back(leaf); v.push_back((uint32_t)st); v.push_back(ridk); v.push_back(qbase);
      v.push_back(mroot); v.push_back(mfin); v.push_back(mzeta); v.push_back(maf); v.push_back(mbeta);
      for (int i = 0; i < 16; ++i) v.push_back(d[1 + i]);
      for (int i = 26; i < (int)kTrRecWords; ++i) v.push_back(0u);
      if (d[0]) {
        uint32_t el[12] = {leaf, (uint32_t)st, flags, ridk};
        for (int i = 0; i < 8; ++i) el[4 + i] = d[1 + i];
        log_pub_tuple(log, BUS_TBLK, true, 1, el, 12);
      } else {
        const uint32_t el[4] = {leaf, (uint32_t)st, flags, qbase};
        log_pub_tuple(log, BUS_TSQ, true, 1, el, 4);
      }
    }
    {
      // the preprocessed commitment is the key's, not the transcript's: the verifier hands it to the runs that end there
      uint32_t el[9] = {leaf_rid(leaf, 0)};
      for (int i = 0; i < 8; ++i) el[1 + i] = vk.prep_root[i];
      log_pub_tuple(log, BUS_ROOT, true, num_queries, el, 9);
      const uint32_t lk[3] = {leaf, (uint32_t)lm - 1, fp_root_of_unity(lm + 1).inv().to_canonical()};
      log_pub_tuple(log, BUS_LEAFK, true, num_queries, lk, 3);
      for (int k = 0; k < lm; ++k) {
        const int lh = lm - k;
        if (!height_present(lh)) continue;
        Fp4 B1 = Fp4::zero(), B2 = Fp4::zero();
        for (int c = 0; c < kNumChips; ++c)
          if (logh[c] == lh) { B1 += b1[c]; B2 += b2[c]; }
        uint32_t el[12] = {leaf, (uint32_t)k, height_has_prep(lh) ? 1u : 0u, fp_root_of_unity(lh).to_canonical()};
        canon4(B1, el + 4);
        canon4(B2, el + 8);
        log_pub_tuple(log, BUS_BCONST, true, num_queries, el, 12);
      }
      const uint32_t pw[2] = {leaf, pow_word};
      log_pub_tuple(log, BUS_POW, false, 1, pw, 2);
    }
  }
  if (stub) return 0;

  // One query's state while it is checked.  Two queries are checked in lockstep (check_queries<2>): their hashing pairs up
  // permutation by permutation, the arithmetic between the hashes is done lane by lane.
  struct QLane {
    uint32_t qi = 0, word = 0;
    LeafCheckLog* log = nullptr;
    std::vector<std::vector<Fp>> rows[4];
    const uint32_t* q = nullptr;
    size_t cs = 0, m = 0;
    Fp4 hsum[4][32];  // with a log: the Horner sums (in alpha_f) of the opened rows, per tree and height
    std::vector<std::array<uint32_t, kQrRecWords>> qrow;
    Fp4 expect, ro_k;
    Fp shift_k;
  };
  const AlphaPows af_pows(af);
  const Fp omi = fp_root_of_unity(lm + 1).inv(), gi0 = Fp::from_canonical(kGenInv);
  const Fp4 d2x = delta * delta, d3x = d2x * delta, d4x = d2x * d2x;
  // reduced openings per height
  auto reduced = [&](const QLane& a, int lh) -> Fp4 {
    Fp4 gsum = Fp4::zero();
    for (int c = 0; c < kNumChips; ++c) {
      if (logh[c] != lh) continue;
      const size_t h = (size_t)1 << lh, mm = a.m & (h - 1);
      const Fp wh = fp_root_of_unity(lh), w2h = fp_root_of_unity(lh + 1);
      const Fp x = (a.cs ? g * w2h : g) * wh.pow(mm);
      Fp4 s1 = Fp4::zero(), s2 = Fp4::zero();
      size_t i = 0, jj = 0;
      for (int r = 0; r < 4; ++r)
        for (int col = 0; col < shape[r].width[c]; ++col, ++i) {
          const Fp v = a.rows[r][c][col];
          s1 += afpow[open_off[c] + i] * v;
          if (r == 1 || r == 2) { s2 += afpow[open_off[c] + n1[c] + jj] * v; ++jj; }
        }
      const Fp4 d0 = (Fp4::from_base(x) - zeta).inv(), d1 = (Fp4::from_base(x) - zeta * wh).inv();
      gsum += (s1 - b1[c]) * d0 + (s2 - b2[c]) * d1;
    }
    return gsum;
  };
  // the query chip's 31 rows (air_machine.hpp), bit 30 of the index word down to bit 0: what does not depend on the openings
  auto start_qrows = [&](QLane& a) {
    a.qrow.resize(31);
    const uint32_t word = a.word, qi = a.qi;
    const size_t cs = a.cs;
    uint32_t eqv = 0, cnt0 = 0, key0 = 0, m0 = 0, keyj = 0, mj = 0;
    // (MT, MT0 and YT are the same on every row of the chain: filled in at the end)
    Fp r_acc = Fp::one(), yki = Fp::zero(), gi = Fp::zero();
    for (int j = 30; j >= 0; --j) {
      std::array<uint32_t, kQrRecWords>& w = a.qrow[30 - j];
      w.fill(0);
      const uint32_t bit = (word >> j) & 1u;
      const bool is_csr = j == lm, is_lay = j < lm, is_fl = j == lm - 1;
      const int k = lm - 1 - j;
      w[QR_IS_REAL] = 1; w[QR_FIRST] = j == 30; w[QR_LAST] = j == 0; w[QR_LEAF] = leaf; w[QR_QL] = qi; w[QR_J] = (uint32_t)j;
      w[QR_BIT] = bit; w[QR_ACC] = word >> j;
      eqv = j == 30 ? bit : (j >= 27 ? (eqv & bit) : eqv);
      w[QR_EQ] = eqv;
      w[QR_F1] = j == 29; w[QR_F2] = j == 28; w[QR_F3] = j == 27;
      w[QR_CSR] = is_csr; w[QR_FL] = is_fl; w[QR_LAY] = is_lay; w[QR_K] = is_lay ? (uint32_t)k : 0u;
      w[QR_CS] = j <= lm ? (uint32_t)cs : 0u;
      w[QR_POW] = 1u << j; w[QR_LOW] = word & ((1u << j) - 1);
      w[QR_REV] = j ? bitrev32(word & ((1u << j) - 1), j) : 0u;
      w[QR_PR0] = j == kTableLogH;
      cnt0 += w[QR_PR0];
      w[QR_CNT0] = cnt0;
      w[QR_P0A] = j < kTableLogH;
      if (j == kTableLogH - 1) { key0 = 1; m0 = 0; }
      if (j < kTableLogH - 1) { key0 = 2 * key0 + ((word >> (j + 1)) & 1u); m0 = 2 * m0 + (height_has_prep(j + 1) ? 1u : 0u); }
      if (j < kTableLogH) { w[QR_KEY0] = key0; w[QR_M0] = m0; w[QR_HAS0] = height_has_prep(j + 1); }
      if (is_fl) { keyj = 1; mj = 0; }
      if (is_lay && !is_fl) { keyj = 2 * keyj + ((word >> (j + 1)) & 1u); mj = 2 * mj + (height_present(j + 1) ? 1u : 0u); }
      if (is_lay) { w[QR_KEYJ] = keyj; w[QR_MJ] = mj; w[QR_HASRO] = height_present(j + 1); }
      w[QR_OMI] = omi.to_canonical();
      w[QR_MU] = bit ? omi.to_canonical() : 1u;
      w[QR_CSM] = w[QR_CS] ? omi.to_canonical() : 1u;
      if (is_lay) {
        r_acc = is_fl ? Fp::from_canonical(w[QR_MU]) : r_acc * r_acc * Fp::from_canonical(w[QR_MU]);
        w[QR_R] = r_acc.to_canonical();
        w[QR_R2] = (r_acc * r_acc).to_canonical();
      }
    }
    // the inverse of y = omega^(cs + 2 m), its squares down the layers, the shifts
    const Fp yt = r_acc * r_acc * (cs ? omi : Fp::one());
    for (int j = 30; j >= 0; --j) {
      std::array<uint32_t, kQrRecWords>& w = a.qrow[30 - j];
      w[QR_YT] = yt.to_canonical();
      w[QR_MT] = 4 * mj;
      w[QR_MT0] = 4 * m0;
      if (j < lm) {
        yki = j == lm - 1 ? yt : yki * yki;
        gi = j == lm - 1 ? gi0 : gi * gi;
        w[QR_YKI] = yki.to_canonical();
        w[QR_GI] = gi.to_canonical();
        const Fp xinv = gi * yki * (((word >> j) & 1u) ? -Fp::one() : Fp::one());
        w[QR_XINV] = xinv.to_canonical();
      }
    }
  };
  // L queries of this proof, checked in lockstep.  An error is the pair's: the caller checks the two one by one to name the query.
  auto check_queries = [&](auto lanes, const uint32_t* qis, LeafCheckLog* const* logs, std::string* err) -> int {
    constexpr int L = decltype(lanes)::value;
    QLane ln[L];
    LeafCheckLog* lg[L];
    for (int t = 0; t < L; ++t) {
      QLane& a = ln[t];
      a.qi = qis[t];
      a.log = lg[t] = logs[t];
      a.q = p_queries + perq * a.qi;
      a.word = index_words[a.qi];
      const size_t idx = indices[a.qi];
      a.cs = idx >> lm;
      a.m = idx & (hmax - 1);
      for (int r = 0; r < 4; ++r) a.rows[r].resize(kNumChips);
    }
    for (int r = 0; r < 4; ++r) {
      const std::vector<std::vector<Fp>>* rowsp[L];
      const uint32_t* pathp[L];
      size_t csv[L], mv[L];
      uint32_t tags[L];
      Fp4* hs[L];
      for (int t = 0; t < L; ++t) {
        QLane& a = ln[t];
        for (int c = 0; c < kNumChips; ++c) {
          a.rows[r][c].resize(shape[r].width[c]);
          for (int i = 0; i < shape[r].width[c]; ++i) a.rows[r][c][i] = Fp::from_canonical(a.q[i]);
          a.q += shape[r].width[c];
        }
        for (auto& hv : a.hsum[r]) hv = Fp4::zero();
        rowsp[t] = &a.rows[r]; pathp[t] = a.q; csv[t] = a.cs; mv[t] = a.m;
        tags[t] = leaf_tag(leaf, a.qi, (uint32_t)r);
        hs[t] = a.log ? a.hsum[r] : nullptr;
      }
      if (!mmcs_verify<L>(shape[r], logh, rowsp, csv, mv, pathp, root[r], kc, lg, tags, leaf_rid(leaf, (uint32_t)r), log ? &af_pows : nullptr,
                          hs)) {
        static const char* names[4] = {"preprocessed", "main", "permutation", "quotient"};
        *err = std::string(names[r]) + " Merkle opening rejected";
        return 8;
      }
      for (int t = 0; t < L; ++t) ln[t].q += 8 * ((size_t)shape[r].lm + 1);
    }
    for (int t = 0; t < L; ++t) {
      QLane& a = ln[t];
      if (a.log) start_qrows(a);
      a.expect = reduced(a, lm);
      a.ro_k = a.expect;  // the reduced opening that joins on the row of layer k (layer 0: the tallest height's)
      a.shift_k = g;
    }
    for (int k = 0; k < lm; ++k) {
      const int loghk = lm - k;
      const size_t hk = (size_t)1 << loghk, half = hk >> 1;
      Fp pair[L][8], cur[L][8];
      const Fp* pairp[L];
      Fp4 lo[L], hi[L];
      size_t leaf_pos[L], mlo[L];
      uint32_t tags[L], key[L];
      for (int t = 0; t < L; ++t) {
        QLane& a = ln[t];
        const size_t mk = a.m & (hk - 1);
        mlo[t] = mk & (half - 1);
        lo[t] = read_fp4(a.q); hi[t] = read_fp4(a.q + 4);
        if ((mk >= half ? hi[t] : lo[t]) != a.expect) { *err = "FRI layer value inconsistent with previous fold"; return 8; }
        for (int i = 0; i < 4; ++i) { pair[t][i] = lo[t].c[i]; pair[t][4 + i] = hi[t].c[i]; }
        pairp[t] = pair[t];
        tags[t] = leaf_tag(leaf, a.qi, 4 + (uint32_t)k);
        key[t] = 1;
        leaf_pos[t] = a.cs * half + mlo[t];
      }
      // the layer's opening: the pair's hash, hashed up to the layer's root (with a log: a run of the Poseidon2 chip)
      sponge_logged<L>(pairp, 8, cur, kc, lg, tags, key, 0, true, false, true);
      for (int l = 0; l < loghk; ++l) {
        Fp sib[L][8], nxt[L][8];
        bool right[L];
        uint32_t kind[L];
        for (int t = 0; t < L; ++t) {
          for (int i = 0; i < 8; ++i) sib[t][i] = Fp::from_canonical(ln[t].q[8 + 8 * l + i]);
          right[t] = (leaf_pos[t] >> l) & 1;
          key[t] = 2 * key[t] + (right[t] ? 1u : 0u);
          kind[t] = right[t] ? P2K_PR : P2K_PL;
        }
        compress_logged<L>(cur, sib, right, nxt, kc, lg, kind, tags, key, 0);
        for (int t = 0; t < L; ++t)
          for (int i = 0; i < 8; ++i) cur[t][i] = nxt[t][i];
      }
      for (int t = 0; t < L; ++t) {
        QLane& a = ln[t];
        for (int i = 0; i < 8; ++i)
          if (cur[t][i] != fri_roots[k][i]) { *err = "FRI Merkle path rejected"; return 8; }
        const Fp xk = (a.cs ? a.shift_k * fp_root_of_unity(loghk + 1) : a.shift_k) * fp_root_of_unity(loghk).pow(mlo[t]);
        const Fp xinv = xk.inv();
        const Fp4 folded = (lo[t] + hi[t]) * inv2 + betas[k] * ((lo[t] - hi[t]) * (inv2 * xinv));
        if (a.log) {
          log_run_end(a.log, leaf_rid(leaf, 4 + (uint32_t)k));
          std::array<uint32_t, kQrRecWords>& w = a.qrow[30 - (lm - 1 - k)];
          if (w[QR_XINV] != xinv.to_canonical() || w[QR_POW] * 2 + w[QR_REV] * 2 + w[QR_CS] != key[t]) { *err = "internal: query chip row"; return 7; }
          canon4(betas[k], &w[QR_BETA]); canon4(lo[t], &w[QR_LO]); canon4(hi[t], &w[QR_HI]); canon4(a.expect, &w[QR_E]); canon4(folded, &w[QR_F]);
          // the reduced opening of the height 2^loghk, which joined on THIS row (ro_k), from its parts
          const Fp yki = Fp::from_canonical(w[QR_YKI]);
          const Fp wh = fp_root_of_unity(loghk);
          const Fp4 zw = zeta * wh;
          Fp4 den0 = zeta * (-yki), den1 = zw * (-yki);
          den0.c[0] = den0.c[0] + g;
          den1.c[0] = den1.c[0] + g;
          const Fp4 d0 = den0.inv() * yki, d1 = den1.inv() * yki;
          canon4(d0, &w[QR_D0]); canon4(d1, &w[QR_D1]);
          if (w[QR_HASRO]) {
            Fp4 B1 = Fp4::zero(), B2 = Fp4::zero();
            for (int c = 0; c < kNumChips; ++c)
              if (logh[c] == loghk) { B1 += b1[c]; B2 += b2[c]; }
            const Fp4 g2 = a.hsum[1][loghk] + delta * a.hsum[2][loghk];
            for (int r = 0; r < 4; ++r) canon4(a.hsum[r][loghk], &w[QR_H + 4 * r]);
            canon4(af, &w[QR_AF]); canon4(delta, &w[QR_DL]); canon4(d2x, &w[QR_D2]); canon4(d3x, &w[QR_D3]); canon4(d4x, &w[QR_D4]);
            canon4(g2, &w[QR_G2]); canon4(zeta, &w[QR_ZETA]); canon4(zw, &w[QR_ZW]); canon4(B1, &w[QR_B1]); canon4(B2, &w[QR_B2]);
            w[QR_WH] = wh.to_canonical();
            const Fp4 ro = d0 * (a.hsum[0][loghk] + delta * a.hsum[1][loghk] + d2x * a.hsum[2][loghk] + d3x * a.hsum[3][loghk] - B1) +
                           d1 * (d4x * g2 - B2);
            if (ro != a.ro_k) { *err = "internal: reduced opening from the Horner sums"; return 7; }
            canon4(ro, &w[QR_RO]);
          } else {
            // (no height joins: ZETA = 0 in the row, so D0 = D1 = 1 / (g y))
            const Fp4 dz = Fp4::from_base(yki * gi0);
            canon4(dz, &w[QR_D0]); canon4(dz, &w[QR_D1]);
          }
        }
        a.expect = folded;
        const bool joins = height_present(loghk - 1);
        a.ro_k = Fp4::zero();
        if (joins) { a.ro_k = reduced(a, loghk - 1); a.expect += a.ro_k; }
        a.q += 8 + 8 * loghk;
        a.shift_k = a.shift_k * a.shift_k;
      }
    }
    for (int t = 0; t < L; ++t) {
      QLane& a = ln[t];
      if (a.expect != final_poly) { *err = "FRI final value mismatch"; return 8; }
      if (a.log)
        for (const auto& w : a.qrow) a.log->qr_rows.insert(a.log->qr_rows.end(), w.begin(), w.end());
    }
    return 0;
  };
  std::vector<int> q_rc(num_queries, 0);
  std::vector<std::string> q_err(num_queries);
  std::vector<LeafCheckLog> q_log(log ? num_queries : 0);
  for (LeafCheckLog& ql : q_log) ql.leaf_index = log->leaf_index;
  std::atomic<uint32_t> next_q{0};
  // (a query that throws - out of memory in its vectors - is a malformed-input failure of that query, on whichever thread)
  auto check_one = [&](uint32_t qi) noexcept {
    try {
      if (log) { q_log[qi].p2_rows.clear(); q_log[qi].qr_rows.clear(); }
      LeafCheckLog* one[1] = {log ? &q_log[qi] : nullptr};
      q_rc[qi] = check_queries(std::integral_constant<int, 1>(), &qi, one, &q_err[qi]);
    } catch (...) {
      q_rc[qi] = 7;
      try { q_err[qi] = "query check ran out of memory"; } catch (...) {}
    }
  };
  const uint32_t step = (uint32_t)lockstep_lanes();  // queries checked side by side: what fills the host's vector unit
  auto check_group = [&](auto lanes, uint32_t q0) noexcept {
    constexpr int L = decltype(lanes)::value;
    int rc = 7;
    try {
      uint32_t qs[L];
      LeafCheckLog* logs[L];
      for (int t = 0; t < L; ++t) { qs[t] = q0 + (uint32_t)t; logs[t] = log ? &q_log[q0 + t] : nullptr; }
      std::string e;
      rc = check_queries(lanes, qs, logs, &e);
    } catch (...) {
    }
    if (rc)  // one of them fails: each on its own, so that the failure has a name
      for (int t = 0; t < L; ++t) check_one(q0 + (uint32_t)t);
  };
  auto worker = [&]() noexcept {
    for (uint32_t qi; (qi = next_q.fetch_add(step)) < num_queries;) {
      uint32_t left = std::min(step, num_queries - qi);
      if (left == 4) { check_group(std::integral_constant<int, 4>(), qi); continue; }
      for (; left >= 2; left -= 2, qi += 2) check_group(std::integral_constant<int, 2>(), qi);
      if (left) check_one(qi);
    }
  };
  {
    struct Joiner {  // joins on every way out of the block
      std::vector<std::thread> th;
      ~Joiner() { for (auto& t : th) if (t.joinable()) t.join(); }
    } pool;
    try {
      for (unsigned t = 1; t < n_thr; ++t) pool.th.emplace_back(worker);
    } catch (...) {
    }  // fewer threads than wanted: the queries are claimed from one counter
    worker();
  }
  for (uint32_t qi = 0; qi < num_queries; ++qi)
    if (q_rc[qi]) { *err = q_err[qi]; return q_rc[qi]; }
  if (log) {
    // the queries' rows in query order: one allocation, copied side by side (11 MB of Poseidon2-chip records per leaf)
    std::vector<size_t> at_p2(num_queries + 1), at_qr(num_queries + 1);
    at_p2[0] = log->p2_rows.size();
    at_qr[0] = log->qr_rows.size();
    for (uint32_t qi = 0; qi < num_queries; ++qi) {
      at_p2[qi + 1] = at_p2[qi] + q_log[qi].p2_rows.size();
      at_qr[qi + 1] = at_qr[qi] + q_log[qi].qr_rows.size();
    }
    log->p2_rows.resize(at_p2[num_queries]);
    log->qr_rows.resize(at_qr[num_queries]);
    std::atomic<uint32_t> next_c{0};
    auto copier = [&]() noexcept {
      for (uint32_t qi; (qi = next_c.fetch_add(1)) < num_queries;) {
        if (!q_log[qi].p2_rows.empty()) memcpy(log->p2_rows.data() + at_p2[qi], q_log[qi].p2_rows.data(), q_log[qi].p2_rows.size() * 4);
        if (!q_log[qi].qr_rows.empty()) memcpy(log->qr_rows.data() + at_qr[qi], q_log[qi].qr_rows.data(), q_log[qi].qr_rows.size() * 4);
        RowWords().swap(q_log[qi].p2_rows);
      }
    };
    struct Joiner {
      std::vector<std::thread> th;
      ~Joiner() { for (auto& t : th) if (t.joinable()) t.join(); }
    } pool;
    try {
      for (unsigned t = 1; t < std::min(n_thr, 4u); ++t) pool.th.emplace_back(copier);
    } catch (...) {
    }
    copier();
  }
  return 0;
}

}  // namespace zksp
