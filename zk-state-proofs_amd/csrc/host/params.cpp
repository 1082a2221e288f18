// Proof-system parameters that need no GPU: the Poseidon2 instance, the per-height
// domain tables and the proof-size formulas.  Kept free of HIP runtime calls so the
// verifier and executor can be built and sanitised on a CPU-only toolchain
// (tests/host_fuzz.cpp).
#include <cstring>

#include "context.hpp"

namespace zksp {

void build_p2_consts(P2Consts* out) {
  static const char tag[] = "zksp/poseidon2/babybear/w16/v1";
  // SHAKE256: rate 136 bytes, domain suffix 0x1f
  uint64_t st[25];
  uint8_t blk[136];
  memset(st, 0, sizeof st);
  memset(blk, 0, sizeof blk);
  size_t n = strlen(tag);
  memcpy(blk, tag, n);
  blk[n] ^= 0x1f;
  blk[135] ^= 0x80;
  for (int i = 0; i < 17; ++i) {
    uint64_t w;
    memcpy(&w, blk + 8 * i, 8);
    st[i] ^= w;
  }
  keccak_f1600(st);
  uint32_t vals[141];
  int got = 0, pos = 0;
  while (got < 141) {
    if (pos == 136) {
      keccak_f1600(st);
      pos = 0;
    }
    uint32_t w;
    memcpy(&w, reinterpret_cast<const uint8_t*>(st) + pos, 4);
    pos += 4;
    w &= 0x7fffffffu;
    if (w < kP) vals[got++] = w;
  }
  int k = 0;
  for (int r = 0; r < 4; ++r)
    for (int i = 0; i < 16; ++i) out->ext[r][i] = Fp::from_canonical(vals[k++]).v;
  for (int r = 0; r < 13; ++r) out->internal[r] = Fp::from_canonical(vals[k++]).v;
  for (int r = 4; r < 8; ++r)
    for (int i = 0; i < 16; ++i) out->ext[r][i] = Fp::from_canonical(vals[k++]).v;
  out->diag[0] = Fp::from_canonical(kP - 2).v;
  for (int i = 1; i < 15; ++i) out->diag[i] = Fp::from_canonical(1u << (i - 1)).v;
  out->diag[15] = Fp::from_canonical(1u << 15).v;
  // derived tables of the signed lazy permutation (poseidon2.hpp): constants times R^2,
  // attached to the linear layer that precedes their round
  auto times_r = [](uint32_t monty) { return (int64_t)(((uint64_t)monty * kRModP) % kP); };
  for (int i = 0; i < 16; ++i) out->sdiag[i] = p2s_centre(out->diag[i]);
  for (int l = 0; l < 9; ++l)
    for (int i = 0; i < 16; ++i) {
      uint32_t v = 0;
      if (l == 4) v = i == 0 ? out->internal[0] : 0;  // external round 3 -> internal round 0
      else if (l < 8) v = out->ext[l][i];              // layer l precedes external round l
      out->lin_rc[l][i] = v;                           // layer 8 closes the permutation
      out->lin_add[l][i] = v ? times_r(v) : 0;         // R^2-scaled form for the cooperative variant
    }
  for (int r = 0; r < 13; ++r) out->int_add[r] = r < 12 ? times_r(out->internal[r + 1]) : 0;
  for (int i = 0; i < 16; ++i) out->int_last[i] = times_r(out->ext[4][i]);
}

const P2Consts& host_p2_consts() {
  static const P2Consts c = [] {
    P2Consts t;
    build_p2_consts(&t);
    return t;
  }();
  return c;
}

static inline uint32_t bitrev32(uint32_t v, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}

void build_host_domain(int logh, HostDomain* d, bool full) {
  const size_t h = (size_t)1 << logh;
  d->logh = logh;
  const Fp g = Fp::from_canonical(kGen);
  const Fp wh = fp_root_of_unity(logh), w2h = fp_root_of_unity(logh + 1);
  const Fp wh_inv = wh.inv();
  d->w_h = wh.v;
  d->tw_fwd.resize(h / 2 ? h / 2 : 1);
  d->tw_inv.resize(h / 2 ? h / 2 : 1);
  Fp a = Fp::one(), b = Fp::one();
  for (size_t i = 0; i < d->tw_fwd.size(); ++i) {
    d->tw_fwd[i] = a.v;
    d->tw_inv[i] = b.v;
    a = a * wh;
    b = b * wh_inv;
  }
  // per-stage copies: the butterflies of stage t read consecutive words instead of a stride of 2^(logh-t)
  d->twc_fwd.assign(h > 1 ? h : 2, Fp::one().v);
  d->twc_inv.assign(h > 1 ? h : 2, Fp::one().v);
  for (int t = 1; t <= logh; ++t)
    for (size_t j = 0; j < ((size_t)1 << (t - 1)); ++j) {
      d->twc_fwd[((size_t)1 << (t - 1)) + j] = d->tw_fwd[j << (logh - t)];
      d->twc_inv[((size_t)1 << (t - 1)) + j] = d->tw_inv[j << (logh - t)];
    }
  const Fp shifts[2] = {g, g * w2h};
  const Fp hinv = Fp::from_canonical((uint32_t)(h % kP)).inv();
  const Fp in_shifts[3] = {Fp::one(), shifts[0], shifts[1]};
  for (int t = 0; t < 3; ++t) {
    d->in_scale_br[t].resize(h);
    const Fp si = in_shifts[t].inv();
    Fp p = hinv;
    std::vector<uint32_t> nat(h);
    for (size_t k = 0; k < h; ++k) {
      nat[k] = p.v;
      p = p * si;
    }
    for (size_t pos = 0; pos < h; ++pos) d->in_scale_br[t][pos] = nat[bitrev32((uint32_t)pos, logh)];
  }
  d->out_scale_br.resize(2 * h);
  if (full) {
    d->xs.resize(2 * h);
    d->sel_first.resize(2 * h);
    d->sel_trans.resize(2 * h);
    d->sel_last.resize(2 * h);
  }
  for (int c = 0; c < 2; ++c) {
    std::vector<uint32_t> nat(h);
    Fp p = Fp::one();
    for (size_t k = 0; k < h; ++k) {
      nat[k] = p.v;
      p = p * shifts[c];
    }
    for (size_t pos = 0; pos < h; ++pos) d->out_scale_br[c * h + pos] = nat[bitrev32((uint32_t)pos, logh)];
    const Fp zh = shifts[c].pow(h) - Fp::one();
    d->zh_inv[c] = zh.inv().v;
    Fp x = shifts[c];
    for (size_t m = 0; full && m < h; ++m) {
      d->xs[c * h + m] = x.v;
      d->sel_first[c * h + m] = (zh * (x - Fp::one()).inv()).v;
      d->sel_trans[c * h + m] = (x - wh_inv).v;
      d->sel_last[c * h + m] = (zh * (x - wh_inv).inv()).v;
      x = x * wh;
    }
  }
}

size_t proof_body_words(int logh, uint32_t num_queries) {
  const size_t logn = (size_t)logh + 1, W = 2633, PW = 4;
  // trace root, running-sum root, cumulative sum, quotient root, opened values, FRI roots, final, witness
  size_t words = 8 + 8 + 4 + 8 + (2 * W + 8 + 2 * PW) * 4 + 8 * (size_t)logh + 4 + 1;
  size_t perq = W + 8 * logn + PW + 8 * logn + 8 + 8 * logn;
  for (int k = 0; k < logh; ++k) perq += 8 + 8 * (size_t)(logh - k);
  return words + perq * num_queries;
}

// 30 fixed words, the public values, then the public I/O list (50 u64 per permutation)
size_t proof_header_words(uint32_t pv_len, uint32_t n_perms) { return 30 + (pv_len + 3) / 4 + (size_t)100 * n_perms; }

// rows of the column-major [8][R] matrix that carries the I/O limbs of a trace height
int bus_io_log_rows(int logh) {
  const size_t max_perms = ((size_t)1 << logh) / 24;
  const size_t rows = (max_perms * 200 + 7) / 8;
  int l = 0;
  while (((size_t)1 << l) < rows) ++l;
  return l;
}

}  // namespace zksp
