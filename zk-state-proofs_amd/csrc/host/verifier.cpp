// Host verifier: replaces `client.verify(&proof, &vk)` (reference
// prover/src/bin/main.rs:80; sp1-stark 3.4.0 + p3-uni-stark / p3-fri verifiers,
// Cargo.lock:7485, :5378, :5253).  Re-derives the Fiat-Shamir transcript, recomputes
// the LogUp bus sum from the public I/O list, checks the constraint identity at zeta
// with the SAME AIR template the device quotient kernel instantiates (air_keccak.hpp),
// then the FRI queries.  Needs no GPU.
#include "verifier.hpp"

#include <cstring>

#include "../device/air_keccak.hpp"
#include "../device/poseidon2.hpp"
#include "executor.hpp"
#include "host_hash.hpp"

namespace zksp {

namespace {

using namespace hosthash;

struct VerifyCtx {
  using F = Fp4;
  using E = Fp4;
  const Fp4* loc;
  const Fp4* nxt;
  Fp4 first, trans, last;
  const Fp4* ap;
  Fp4 acc;
  // bus
  Fp4 gamma_, cum_, phi_, phi_next_;
  const Fp4* bpow;
  F local(int col) const { return loc[col]; }
  F next(int col) const { return nxt[col]; }
  F is_first() const { return first; }
  F is_trans() const { return trans; }
  F is_last() const { return last; }
  F one() const { return Fp4::one(); }
  void emit_at(int k, F v) { acc += ap[k] * v; }
  E gamma() const { return gamma_; }
  E beta_pow(int j) const { return bpow[j]; }
  E cum_sum() const { return cum_; }
  E phi_local() const { return phi_; }
  E phi_next() const { return phi_next_; }
  E lift(F v) const { return v; }
  void emit_ext_at(int k, E v) { acc += ap[k] * v; }
};

bool all_canonical(const uint32_t* w, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (w[i] >= kP) return false;
  return true;
}

Fp4 read_fp4(const uint32_t* w) {
  Fp4 r;
  for (int i = 0; i < 4; ++i) r.c[i] = Fp::from_canonical(w[i]);
  return r;
}

int ceil_log2(size_t v) {
  int l = 0;
  while (((size_t)1 << l) < v) ++l;
  return l;
}

}  // namespace

bool parse_proof_header(const uint8_t* bytes, size_t len, ProofHeader* h, std::string* err) {
  if (len < 30 * 4 || (len & 3)) { *err = "proof too short"; return false; }
  const uint32_t* w = reinterpret_cast<const uint32_t*>(bytes);
  if (w[0] != kProofMagic) { *err = "bad magic"; return false; }
  if (w[1] != kProofVersion) { *err = "unsupported proof version"; return false; }
  h->log_h = w[2];
  h->n_perms = w[3];
  h->exit_code = w[4];
  h->pv_len = w[5];
  memcpy(h->pv_digest, w + 6, 32);
  memcpy(h->deferred_digest, w + 14, 32);
  memcpy(h->vk_digest, w + 22, 32);
  // the prover emits 2^5..2^14 (load_batch); a larger claimed height would only make the verifier
  // hash a huge zero-padded I/O matrix before any check can reject the proof
  if (h->log_h < 5 || h->log_h > 14) { *err = "log_h out of range"; return false; }
  if (h->pv_len > (1u << 24)) { *err = "public values too long"; return false; }
  if ((uint64_t)h->n_perms * 24 > ((uint64_t)1 << h->log_h)) { *err = "n_perms exceeds trace height"; return false; }
  size_t hw = proof_header_words(h->pv_len, h->n_perms);
  if (len < hw * 4) { *err = "proof truncated in header"; return false; }
  h->pv_offset = 30 * 4;
  h->io_offset = (30 + (h->pv_len + 3) / 4) * 4;
  h->body_offset = hw * 4;
  return true;
}

int verify_proof(const uint8_t* bytes, size_t len, const uint32_t vk_digest[8], uint32_t num_queries, uint32_t pow_bits,
                 std::string* err) {
  ProofHeader hd;
  if (!parse_proof_header(bytes, len, &hd, err)) return 7;
  const int logh = (int)hd.log_h, logn = logh + 1, W = ka::kWidth, PW = ka::kPermWidth;
  const size_t h = (size_t)1 << logh;
  if (len != hd.body_offset + proof_body_words(logh, num_queries) * 4) { *err = "proof length mismatch"; return 7; }
  if (memcmp(hd.vk_digest, vk_digest, 32) != 0) { *err = "verifying key mismatch"; return 8; }
  // a guest that panicked has no proof in the reference either (run() fails, main.rs:71-74)
  if (hd.exit_code != 0) { *err = "guest exit code is not zero"; return 8; }
  // the guest commits sha256(public values) word by word
  {
    uint8_t dg[32];
    sha256(bytes + hd.pv_offset, hd.pv_len, dg);
    if (memcmp(dg, hd.pv_digest, 32) != 0) { *err = "public-values digest mismatch"; return 8; }
  }
  const uint32_t* body = reinterpret_cast<const uint32_t*>(bytes + hd.body_offset);
  const size_t body_words = proof_body_words(logh, num_queries);
  if (!all_canonical(body, body_words)) { *err = "non-canonical field element"; return 7; }

  const P2Consts* kc = &host_p2_consts();
  const size_t n_open = (size_t)(2 * W + 8 + 2 * PW);
  const uint32_t* p_root_t = body;
  const uint32_t* p_root_p = body + 8;
  const uint32_t* p_cum = body + 16;
  const uint32_t* p_root_q = body + 20;
  const uint32_t* p_opened = body + 28;
  const uint32_t* p_fri_roots = p_opened + n_open * 4;
  const uint32_t* p_final = p_fri_roots + 8 * (size_t)logh;
  const uint32_t* p_witness = p_final + 4;
  const uint32_t* p_queries = p_witness + 1;

  // public I/O list -> limbs (input 100, output 100 per permutation)
  std::vector<Fp> io_limbs((size_t)ka::kBusTuple * hd.n_perms);
  {
    const uint8_t* io = bytes + hd.io_offset;
    for (uint32_t p = 0; p < hd.n_perms; ++p)
      for (int half = 0; half < 2; ++half)
        for (int lane = 0; lane < 25; ++lane) {
          uint64_t v;
          memcpy(&v, io + ((size_t)p * 50 + half * 25 + lane) * 8, 8);
          for (int l = 0; l < 4; ++l)
            io_limbs[(size_t)p * ka::kBusTuple + 100 * half + 4 * lane + l] =
                Fp::from_canonical((uint32_t)((v >> (16 * l)) & 0xffff));
        }
  }

  HostChallenger ch(kc);
  for (int i = 0; i < 8; ++i) ch.observe_canon(hd.vk_digest[i]);
  ch.observe_canon(hd.log_h);
  ch.observe_canon(hd.n_perms);
  ch.observe_canon(hd.exit_code & 0xffff);
  ch.observe_canon(hd.exit_code >> 16);
  for (int i = 0; i < 8; ++i) { ch.observe_canon(hd.pv_digest[i] & 0xffff); ch.observe_canon(hd.pv_digest[i] >> 16); }
  for (int i = 0; i < 8; ++i) { ch.observe_canon(hd.deferred_digest[i] & 0xffff); ch.observe_canon(hd.deferred_digest[i] >> 16); }
  {
    Fp io_root[8];
    list_root(io_limbs, bus_io_log_rows(logh), io_root, kc);
    for (int i = 0; i < 8; ++i) ch.observe(io_root[i]);
  }
  Fp root_t[8], root_p[8], root_q[8];
  for (int i = 0; i < 8; ++i) { root_t[i] = Fp::from_canonical(p_root_t[i]); ch.observe(root_t[i]); }
  const Fp4 gamma = ch.sample_ext();
  const Fp4 beta = ch.sample_ext();
  for (int i = 0; i < 8; ++i) { root_p[i] = Fp::from_canonical(p_root_p[i]); ch.observe(root_p[i]); }
  const Fp4 cum_sum = read_fp4(p_cum);
  for (int i = 0; i < 4; ++i) ch.observe(cum_sum.c[i]);
  const Fp4 alpha = ch.sample_ext();
  for (int i = 0; i < 8; ++i) { root_q[i] = Fp::from_canonical(p_root_q[i]); ch.observe(root_q[i]); }
  const Fp4 zeta = ch.sample_ext();

  // ---- LogUp bus: the chip must have received exactly the public list ----
  std::vector<Fp4> bpow(ka::kBusTuple);
  bpow[0] = Fp4::one();
  for (int j = 1; j < ka::kBusTuple; ++j) bpow[j] = bpow[j - 1] * beta;
  {
    Fp4 expect = Fp4::zero();
    for (uint32_t p = 0; p < hd.n_perms; ++p) {
      Fp4 f = gamma;
      for (int j = 0; j < ka::kBusTuple; ++j) f += bpow[j] * io_limbs[(size_t)p * ka::kBusTuple + j];
      expect += f.inv();
    }
    if (expect != cum_sum) { *err = "LogUp bus sum does not match the public I/O list"; return 8; }
  }

  std::vector<Fp4> opened(n_open);
  for (size_t i = 0; i < n_open; ++i) opened[i] = read_fp4(p_opened + 4 * i);
  {
    std::vector<Fp> words(n_open * 4);
    for (size_t t = 0; t < n_open * 4; ++t) words[t] = Fp::from_canonical(p_opened[t]);
    Fp open_root[8];
    list_root(words, ceil_log2((n_open * 4 + 7) / 8), open_root, kc);
    for (int i = 0; i < 8; ++i) ch.observe(open_root[i]);
  }
  const Fp4 af = ch.sample_ext();

  // ---- constraint identity at zeta ----
  const Fp g = Fp::from_canonical(kGen);
  const Fp wh = fp_root_of_unity(logh), w2h = fp_root_of_unity(logn);
  const Fp wh_inv = wh.inv();
  const Fp4 zeta_h = zeta.pow(h);
  const Fp4 zh = zeta_h - Fp4::one();
  const size_t o_pl = (size_t)2 * W + 8, o_pn = o_pl + PW;
  {
    std::vector<Fp4> apow(ka::kNumAllConstraints);
    apow[0] = Fp4::one();
    for (int k = 1; k < ka::kNumAllConstraints; ++k) apow[k] = apow[k - 1] * alpha;
    VerifyCtx vc;
    vc.loc = opened.data();
    vc.nxt = opened.data() + W;
    vc.first = zh * (zeta - Fp4::one()).inv();
    vc.trans = zeta - Fp4::from_base(wh_inv);
    vc.last = zh * (zeta - Fp4::from_base(wh_inv)).inv();
    vc.acc = Fp4::zero();
    vc.ap = apow.data();
    vc.gamma_ = gamma;
    vc.cum_ = cum_sum;
    vc.bpow = bpow.data();
    vc.phi_ = vc.phi_next_ = Fp4::zero();
    for (int j = 0; j < PW; ++j) {
      Fp4 basis = Fp4::zero();
      basis.c[j] = Fp::one();
      vc.phi_ += basis * opened[o_pl + j];
      vc.phi_next_ += basis * opened[o_pn + j];
    }
    for (int task = 0; task < ka::kBusTask; ++task) ka::eval_task(task, vc);
    ka::eval_bus(vc);
    // quotient(zeta) from its two chunk polynomials
    Fp4 q[2];
    for (int c = 0; c < 2; ++c) {
      q[c] = Fp4::zero();
      for (int j = 0; j < 4; ++j) {
        Fp4 basis = Fp4::zero();
        basis.c[j] = Fp::one();
        q[c] += basis * opened[2 * W + 4 * c + j];
      }
    }
    const Fp sh = g.pow(h);  // s^H; (s*w_2H)^H = -s^H
    const Fp inv_2sh = (sh + sh).inv();
    Fp4 quot = q[0] * (zeta_h + Fp4::from_base(sh)) * inv_2sh - q[1] * (zeta_h - Fp4::from_base(sh)) * inv_2sh;
    if (vc.acc != quot * zh) { *err = "constraint identity fails at zeta"; return 8; }
  }

  // ---- FRI transcript ----
  std::vector<Fp4> betas(logh);
  std::vector<std::array<Fp, 8>> fri_roots(logh);
  for (int k = 0; k < logh; ++k) {
    for (int i = 0; i < 8; ++i) { fri_roots[k][i] = Fp::from_canonical(p_fri_roots[8 * k + i]); ch.observe(fri_roots[k][i]); }
    betas[k] = ch.sample_ext();
  }
  const Fp4 final_poly = read_fp4(p_final);
  for (int i = 0; i < 4; ++i) ch.observe(final_poly.c[i]);
  ch.observe_canon(p_witness[0]);
  if (ch.sample_bits((int)pow_bits) != 0) { *err = "proof-of-work witness rejected"; return 8; }

  // ---- reduced-opening constants ----
  std::vector<Fp4> afpow(n_open);
  afpow[0] = Fp4::one();
  for (size_t i = 1; i < n_open; ++i) afpow[i] = afpow[i - 1] * af;
  Fp4 b0 = Fp4::zero(), b1 = Fp4::zero(), b2 = Fp4::zero(), b3 = Fp4::zero(), b4 = Fp4::zero();
  for (int i = 0; i < W; ++i) {
    b0 += afpow[i] * opened[i];
    b1 += afpow[i] * opened[W + i];
  }
  for (int i = 0; i < 8; ++i) b2 += afpow[i] * opened[2 * W + i];
  for (int i = 0; i < PW; ++i) {
    b3 += afpow[i] * opened[o_pl + i];
    b4 += afpow[i] * opened[o_pn + i];
  }
  const Fp4 zeta_next = zeta * wh;
  const Fp inv2 = Fp::from_canonical(2).inv();

  size_t perq = (size_t)W + 8 * (size_t)logn + PW + 8 * (size_t)logn + 8 + 8 * (size_t)logn;
  for (int k = 0; k < logh; ++k) perq += 8 + 8 * (size_t)(logh - k);
  std::vector<Fp> row(W);
  for (uint32_t qi = 0; qi < num_queries; ++qi) {
    const uint32_t* q = p_queries + perq * qi;
    const size_t idx = ch.sample_bits(logn);
    const size_t c = idx >> logh, m = idx & (h - 1);
    // trace row
    for (int i = 0; i < W; ++i) row[i] = Fp::from_canonical(q[i]);
    Fp leaf[8];
    hash_elems(row.data(), W, leaf, kc);
    if (!verify_path(leaf, idx, q + W, logn, root_t, kc)) { *err = "trace Merkle path rejected"; return 8; }
    Fp4 st = Fp4::zero();
    for (int i = 0; i < W; ++i) st += afpow[i] * row[i];
    q += W + 8 * logn;
    // running-sum row
    Fp prow[4];
    for (int i = 0; i < PW; ++i) prow[i] = Fp::from_canonical(q[i]);
    hash_elems(prow, PW, leaf, kc);
    if (!verify_path(leaf, idx, q + PW, logn, root_p, kc)) { *err = "running-sum Merkle path rejected"; return 8; }
    Fp4 sp = Fp4::zero();
    for (int i = 0; i < PW; ++i) sp += afpow[i] * prow[i];
    q += PW + 8 * logn;
    // quotient row
    Fp qrow[8];
    for (int i = 0; i < 8; ++i) qrow[i] = Fp::from_canonical(q[i]);
    hash_elems(qrow, 8, leaf, kc);
    if (!verify_path(leaf, idx, q + 8, logn, root_q, kc)) { *err = "quotient Merkle path rejected"; return 8; }
    Fp4 sq = Fp4::zero();
    for (int i = 0; i < 8; ++i) sq += afpow[i] * qrow[i];
    q += 8 + 8 * logn;
    const Fp shift_c = c ? g * w2h : g;
    const Fp x = shift_c * wh.pow(m);
    const Fp4 d0 = (Fp4::from_base(x) - zeta).inv(), d1 = (Fp4::from_base(x) - zeta_next).inv();
    Fp4 expect = (st - b0) * d0 + afpow[W] * (st - b1) * d1 + afpow[2 * W] * (sq - b2) * d0 +
                 afpow[o_pl] * (sp - b3) * d0 + afpow[o_pn] * (sp - b4) * d1;
    // FRI layers
    Fp shift_k = g;
    for (int k = 0; k < logh; ++k) {
      const int loghk = logh - k;
      const size_t hk = (size_t)1 << loghk, half = hk >> 1;
      const size_t mk = m & (hk - 1), mlo = mk & (half - 1);
      const Fp4 lo = read_fp4(q), hi = read_fp4(q + 4);
      if ((mk >= half ? hi : lo) != expect) { *err = "FRI layer value inconsistent with previous fold"; return 8; }
      Fp pair[8];
      for (int i = 0; i < 4; ++i) { pair[i] = lo.c[i]; pair[4 + i] = hi.c[i]; }
      hash_elems(pair, 8, leaf, kc);
      if (!verify_path(leaf, c * half + mlo, q + 8, loghk, fri_roots[k].data(), kc)) { *err = "FRI Merkle path rejected"; return 8; }
      const Fp xk = (c ? shift_k * fp_root_of_unity(loghk + 1) : shift_k) * fp_root_of_unity(loghk).pow(mlo);
      expect = (lo + hi) * inv2 + betas[k] * ((lo - hi) * (inv2 * xk.inv()));
      q += 8 + 8 * loghk;
      shift_k = shift_k * shift_k;
    }
    if (expect != final_poly) { *err = "FRI final value mismatch"; return 8; }
  }
  return 0;
}

}  // namespace zksp
