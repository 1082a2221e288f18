/* ORACLE -- TEST INFRASTRUCTURE ONLY (see field.h).
 *
 * Whole-proof CPU restatement: commit -> quotient -> open -> FRI -> queries.
 * Restates the published uni-STARK / two-adic FRI flow of p3-uni-stark, p3-fri,
 * p3-commit and p3-challenger 0.1.4-succinct (reference Cargo.lock:5378, :5253,
 * :5211, :5197) as driven by sp1-stark (Cargo.lock:7485) beneath the reference's
 * `client.prove(&pk, stdin).run()` (prover/src/bin/main.rs:71-74).
 * PARITY UNPINNED vs SP1 proof bytes: transcript order, the Merkle-ised opened-
 * value digest, ascending alpha powers and the byte layout are this repository's
 * own format (DESIGN.md "Proof format"); blowup 2, degree-4 extension, 100
 * queries and 16 PoW bits follow SURVEY.md appendix B's recollection of SP1's
 * core configuration.  The HIP prover must reproduce these bytes exactly.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
static double now_s(void){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+1e-9*t.tv_nsec;}
#define TICK(name) do{ if(getenv("ZKSP_ORACLE_TIMING")){double t_=now_s(); fprintf(stderr,"[oracle] %-10s %.3f s\n",name,t_-t0_); t0_=t_;} }while(0)

#include "zksp_oracle.h"

void orc_fri_fold(const uint32_t* in, int loghk, uint32_t shift_k, const uint32_t* beta4, uint32_t* out) {
  size_t hk = (size_t)1 << loghk, half = hk >> 1;
  fe4 beta;
  memcpy(beta.c, beta4, 16);
  fe w2 = f_root_of_unity(loghk + 1), w = f_root_of_unity(loghk);
  fe inv2 = f_inv(2);
  for (int c = 0; c < 2; ++c) {
    fe base = c ? f_mul(shift_k, w2) : shift_k;
    for (size_t m = 0; m < half; ++m) {
      fe4 lo, hi;
      memcpy(lo.c, in + 4 * ((size_t)c * hk + m), 16);
      memcpy(hi.c, in + 4 * ((size_t)c * hk + m + half), 16);
      fe x = f_mul(base, f_pow(w, m));
      fe4 sum = e_mul_base(e_add(lo, hi), inv2);
      fe4 dif = e_mul_base(e_sub(lo, hi), f_mul(inv2, f_inv(x)));
      fe4 r = e_add(sum, e_mul(beta, dif));
      memcpy(out + 4 * ((size_t)c * half + m), r.c, 16);
    }
  }
}

static fe4 eval_poly(const uint32_t* coef, size_t n, fe4 z) {
  fe4 acc = e_zero();
  for (size_t k = n; k-- > 0;) acc = e_add(e_mul(acc, z), e_from(coef[k]));
  return acc;
}

static int ceil_log2(size_t v) {
  int l = 0;
  while (((size_t)1 << l) < v) ++l;
  return l;
}

size_t orc_proof_header_words(uint32_t pv_len, uint32_t n_perms) {
  /* 30 fixed words, public values, then the public I/O list: 50 u64 per permutation */
  return 30 + (pv_len + 3) / 4 + (size_t)100 * n_perms;
}

size_t orc_proof_size(int logh, const orc_config* cfg, uint32_t pv_len, uint32_t n_perms) {
  size_t logn = (size_t)logh + 1;
  size_t words = orc_proof_header_words(pv_len, n_perms);
  /* trace root, perm root, cumulative sum, quotient root, opened values, FRI roots, final, witness */
  words += 8 + 8 + 4 + 8 + (2 * KA_WIDTH + 8 + 2 * KA_PERM_WIDTH) * 4 + 8 * (size_t)logh + 4 + 1;
  size_t perq = KA_WIDTH + 8 * logn + KA_PERM_WIDTH + 8 * logn + 8 + 8 * logn;
  for (int k = 0; k < logh; ++k) perq += 8 + 8 * (size_t)(logh - k);
  words += perq * cfg->num_queries;
  return words * 4;
}

typedef struct {
  uint32_t* w;
  size_t n, cap;
} wbuf;
static void put(wbuf* b, const uint32_t* v, size_t n) {
  if (b->n + n <= b->cap) memcpy(b->w + b->n, v, n * 4);
  b->n += n;
}
static void put_path(wbuf* b, const uint32_t* tree, int logn, size_t idx) {
  for (int l = 0; l < logn; ++l) {
    size_t sib = (idx >> l) ^ 1;
    put(b, tree + 8 * (orc_merkle_layer_offset(logn, l) + sib), 8);
  }
}
static void observe_word_halves(orc_challenger* ch, const uint32_t* w, int n) {
  for (int i = 0; i < n; ++i) {
    orc_ch_observe(ch, w[i] & 0xffff);
    orc_ch_observe(ch, w[i] >> 16);
  }
}
/* Merkle root of a flat word list laid out column-major as [8][2^logr], zero padded:
 * the parallel-friendly way this format absorbs long lists into the transcript */
static void observe_list_root(orc_challenger* ch, const uint32_t* words, size_t n_words, int logr) {
  size_t r = (size_t)1 << logr;
  uint32_t* vpad = (uint32_t*)calloc(8 * r, 4);
  memcpy(vpad, words, n_words * 4);
  uint32_t* tree = (uint32_t*)malloc(8 * (2 * r - 1) * 4);
  orc_merkle_commit(vpad, 8, logr, tree);
  orc_ch_observe_many(ch, tree + 8 * (2 * r - 2), 8);
  free(vpad);
  free(tree);
}

int orc_prove(const uint64_t* states_in, const orc_header* hdr, const uint8_t* public_values, const orc_config* cfg,
              uint8_t* out, size_t cap, size_t* out_len) {
  const int logh = (int)hdr->log_h, logn = logh + 1;
  const size_t h = (size_t)1 << logh, n = h * 2;
  const int W = KA_WIDTH, PW = KA_PERM_WIDTH;
  if (logh < 1 || logh > 26 || 24 * (size_t)hdr->n_perms > h) return 1;
  size_t need = orc_proof_size(logh, cfg, hdr->pv_len, hdr->n_perms);
  *out_len = need;
  if (cap < need) return 2;
  wbuf pb = {(uint32_t*)out, 0, cap / 4};
  double t0_ = now_s();

  /* header: fixed words, public values, public I/O list (input || output state per permutation) */
  uint32_t head[6] = {ZKSP_MAGIC, ZKSP_VERSION, hdr->log_h, hdr->n_perms, hdr->exit_code, hdr->pv_len};
  put(&pb, head, 6);
  put(&pb, hdr->pv_digest, 8);
  put(&pb, hdr->deferred_digest, 8);
  put(&pb, hdr->vk_digest, 8);
  {
    size_t pw = (hdr->pv_len + 3) / 4;
    uint32_t* tmp = (uint32_t*)calloc(pw ? pw : 1, 4);
    memcpy(tmp, public_values, hdr->pv_len);
    put(&pb, tmp, pw);
    free(tmp);
  }
  for (uint32_t p = 0; p < hdr->n_perms; ++p) {
    uint64_t io[50];
    memcpy(io, states_in + 25 * (size_t)p, 200);
    memcpy(io + 25, io, 200);
    orc_keccak_f(io + 25);
    put(&pb, (const uint32_t*)io, 100);
  }
  uint32_t* io_limbs = (uint32_t*)calloc((size_t)KA_BUS_TUPLE * (hdr->n_perms ? hdr->n_perms : 1), 4);
  orc_bus_io_limbs(states_in, (int)hdr->n_perms, io_limbs);

  /* 1. main trace commitment */
  uint32_t* trace = (uint32_t*)malloc((size_t)W * h * 4);
  uint32_t* lde_t = (uint32_t*)malloc((size_t)W * n * 4);
  uint32_t* coef_t = (uint32_t*)malloc((size_t)W * h * 4);
  orc_keccak_trace(states_in, (int)hdr->n_perms, logh, trace);
  orc_coset_lde(trace, logh, W, 1, lde_t, coef_t);
  uint32_t* tree_t = (uint32_t*)malloc(8 * (2 * n - 1) * 4);
  orc_merkle_commit(lde_t, W, logn, tree_t);
  const uint32_t* root_t = tree_t + 8 * (2 * n - 2);

  TICK("commit_t");
  orc_challenger ch;
  orc_ch_init(&ch);
  orc_ch_observe_many(&ch, hdr->vk_digest, 8);
  orc_ch_observe(&ch, hdr->log_h);
  orc_ch_observe(&ch, hdr->n_perms);
  orc_ch_observe(&ch, hdr->exit_code & 0xffff);
  orc_ch_observe(&ch, hdr->exit_code >> 16);
  observe_word_halves(&ch, hdr->pv_digest, 8);
  observe_word_halves(&ch, hdr->deferred_digest, 8);
  observe_list_root(&ch, io_limbs, (size_t)KA_BUS_TUPLE * hdr->n_perms, orc_bus_io_log_rows(logh));
  orc_ch_observe_many(&ch, root_t, 8);
  put(&pb, root_t, 8);

  /* 2. LogUp bus: running-sum trace, committed after gamma and beta are drawn */
  uint32_t gamma[4], beta[4], cum_sum[4];
  orc_ch_sample_ext(&ch, gamma);
  orc_ch_sample_ext(&ch, beta);
  uint32_t* phi = (uint32_t*)malloc((size_t)PW * h * 4);
  orc_bus_perm_trace(trace, logh, gamma, beta, phi, cum_sum);
  free(trace);
  {
    uint32_t expect[4];
    orc_bus_expected_sum(io_limbs, (int)hdr->n_perms, gamma, beta, expect);
    if (memcmp(expect, cum_sum, 16) != 0) return 5; /* the chip did not receive exactly the public list */
  }
  free(io_limbs);
  uint32_t* lde_p = (uint32_t*)malloc((size_t)PW * n * 4);
  uint32_t* coef_p = (uint32_t*)malloc((size_t)PW * h * 4);
  orc_coset_lde(phi, logh, PW, 1, lde_p, coef_p);
  free(phi);
  uint32_t* tree_p = (uint32_t*)malloc(8 * (2 * n - 1) * 4);
  orc_merkle_commit(lde_p, PW, logn, tree_p);
  const uint32_t* root_p = tree_p + 8 * (2 * n - 2);
  orc_ch_observe_many(&ch, root_p, 8);
  orc_ch_observe_many(&ch, cum_sum, 4);
  put(&pb, root_p, 8);
  put(&pb, cum_sum, 4);

  /* 3. quotient */
  uint32_t alpha[4];
  orc_ch_sample_ext(&ch, alpha);
  uint32_t* quot = (uint32_t*)malloc(8 * h * 4);
  orc_keccak_quotient_bus(lde_t, lde_p, logh, alpha, gamma, beta, cum_sum, quot);
  uint32_t* lde_q = (uint32_t*)malloc(8 * n * 4);
  uint32_t* coef_q = (uint32_t*)malloc(8 * h * 4);
  fe w2h = f_root_of_unity(logn), wh = f_root_of_unity(logh);
  for (int c = 0; c < 2; ++c) {
    fe shift_c = c ? f_mul(F_GEN, w2h) : F_GEN;
    orc_coset_lde(quot + (size_t)4 * c * h, logh, 4, shift_c, lde_q + (size_t)4 * c * n, coef_q + (size_t)4 * c * h);
  }
  free(quot);
  uint32_t* tree_q = (uint32_t*)malloc(8 * (2 * n - 1) * 4);
  orc_merkle_commit(lde_q, 8, logn, tree_q);
  const uint32_t* root_q = tree_q + 8 * (2 * n - 2);
  orc_ch_observe_many(&ch, root_q, 8);
  put(&pb, root_q, 8);

  TICK("quotient");
  /* 4. openings at zeta and zeta*w_H: trace (local, next), quotient, running sum (local, next) */
  fe4 zeta, zeta_next;
  orc_ch_sample_ext(&ch, zeta.c);
  zeta_next = e_mul_base(zeta, wh);
  const size_t n_open = (size_t)(2 * W + 8 + 2 * PW);
  fe4* opened = (fe4*)malloc(n_open * sizeof(fe4));
#pragma omp parallel for schedule(static)
  for (int i = 0; i < W; ++i) {
    opened[i] = eval_poly(coef_t + (size_t)i * h, h, zeta);
    opened[W + i] = eval_poly(coef_t + (size_t)i * h, h, zeta_next);
  }
  for (int i = 0; i < 8; ++i) opened[2 * W + i] = eval_poly(coef_q + (size_t)i * h, h, zeta);
  for (int i = 0; i < PW; ++i) {
    opened[2 * W + 8 + i] = eval_poly(coef_p + (size_t)i * h, h, zeta);
    opened[2 * W + 8 + PW + i] = eval_poly(coef_p + (size_t)i * h, h, zeta_next);
  }
  put(&pb, (const uint32_t*)opened, n_open * 4);
  observe_list_root(&ch, (const uint32_t*)opened, n_open * 4, ceil_log2((n_open * 4 + 7) / 8));

  TICK("open");
  /* 5. reduced openings */
  fe4 af;
  orc_ch_sample_ext(&ch, af.c);
  fe4* afpow = (fe4*)malloc(n_open * sizeof(fe4));
  afpow[0] = e_one();
  for (size_t i = 1; i < n_open; ++i) afpow[i] = e_mul(afpow[i - 1], af);
  fe4 b0 = e_zero(), b1 = e_zero(), b2 = e_zero(), b3 = e_zero(), b4 = e_zero();
  for (int i = 0; i < W; ++i) {
    b0 = e_add(b0, e_mul(afpow[i], opened[i]));
    b1 = e_add(b1, e_mul(afpow[i], opened[W + i]));
  }
  for (int i = 0; i < 8; ++i) b2 = e_add(b2, e_mul(afpow[i], opened[2 * W + i]));
  for (int i = 0; i < PW; ++i) {
    b3 = e_add(b3, e_mul(afpow[i], opened[2 * W + 8 + i]));
    b4 = e_add(b4, e_mul(afpow[i], opened[2 * W + 8 + PW + i]));
  }
  uint32_t* layer = (uint32_t*)malloc(n * 16);
  for (int c = 0; c < 2; ++c) {
    fe shift_c = c ? f_mul(F_GEN, w2h) : F_GEN;
#pragma omp parallel for schedule(static)
    for (size_t m = 0; m < h; ++m) {
      fe x = f_mul(shift_c, f_pow(wh, m));
      fe4 st = e_zero(), sq = e_zero(), sp = e_zero();
      for (int i = 0; i < W; ++i) st = e_add(st, e_mul_base(afpow[i], lde_t[((size_t)i * 2 + c) * h + m]));
      for (int i = 0; i < 8; ++i) sq = e_add(sq, e_mul_base(afpow[i], lde_q[((size_t)i * 2 + c) * h + m]));
      for (int i = 0; i < PW; ++i) sp = e_add(sp, e_mul_base(afpow[i], lde_p[((size_t)i * 2 + c) * h + m]));
      fe4 d0 = e_inv(e_sub(e_from(x), zeta));
      fe4 d1 = e_inv(e_sub(e_from(x), zeta_next));
      fe4 g = e_mul(e_sub(st, b0), d0);
      g = e_add(g, e_mul(e_mul(afpow[W], e_sub(st, b1)), d1));
      g = e_add(g, e_mul(e_mul(afpow[2 * W], e_sub(sq, b2)), d0));
      g = e_add(g, e_mul(e_mul(afpow[2 * W + 8], e_sub(sp, b3)), d0));
      g = e_add(g, e_mul(e_mul(afpow[2 * W + 8 + PW], e_sub(sp, b4)), d1));
      memcpy(layer + 4 * ((size_t)c * h + m), g.c, 16);
    }
  }
  free(afpow);
  free(opened);

  TICK("reduce");
  /* 6. FRI commit phase */
  uint32_t** fri_tree = (uint32_t**)malloc((size_t)logh * sizeof(uint32_t*));
  uint32_t** fri_layer = (uint32_t**)malloc((size_t)logh * sizeof(uint32_t*));
  fe shift_k = F_GEN;
  for (int k = 0; k < logh; ++k) {
    int loghk = logh - k;
    size_t hk = (size_t)1 << loghk, half = hk >> 1;
    /* leaf (c, m) = (f[c][m], f[c][m + half]); matrix [8][hk] column-major */
    uint32_t* mat = (uint32_t*)malloc(8 * hk * 4);
    for (int c = 0; c < 2; ++c)
      for (size_t m = 0; m < half; ++m)
        for (int j = 0; j < 4; ++j) {
          mat[(size_t)j * hk + c * half + m] = layer[4 * ((size_t)c * hk + m) + j];
          mat[(size_t)(4 + j) * hk + c * half + m] = layer[4 * ((size_t)c * hk + m + half) + j];
        }
    fri_tree[k] = (uint32_t*)malloc(8 * (2 * hk - 1) * 4);
    orc_merkle_commit(mat, 8, loghk, fri_tree[k]);
    free(mat);
    const uint32_t* root = fri_tree[k] + 8 * (2 * hk - 2);
    orc_ch_observe_many(&ch, root, 8);
    put(&pb, root, 8);
    uint32_t fbeta[4];
    orc_ch_sample_ext(&ch, fbeta);
    uint32_t* nxt = (uint32_t*)malloc(hk * 16);
    orc_fri_fold(layer, loghk, shift_k, fbeta, nxt);
    fri_layer[k] = layer;
    layer = nxt;
    shift_k = f_mul(shift_k, shift_k);
  }
  /* layer now holds f[c][0], c = 0,1: a constant polynomial */
  if (memcmp(layer, layer + 4, 16) != 0) return 3;
  orc_ch_observe_many(&ch, layer, 4);
  put(&pb, layer, 4);
  free(layer);

  TICK("fri");
  /* 7. proof of work, 8. queries */
  uint32_t witness = orc_ch_grind(&ch, (int)cfg->pow_bits);
  put(&pb, &witness, 1);
  TICK("grind");
  uint32_t* row = (uint32_t*)malloc((size_t)W * 4);
  for (uint32_t q = 0; q < cfg->num_queries; ++q) {
    size_t idx = orc_ch_sample_bits(&ch, logn);
    size_t c = idx >> logh, m = idx & (h - 1);
    for (int i = 0; i < W; ++i) row[i] = lde_t[((size_t)i * 2 + c) * h + m];
    put(&pb, row, (size_t)W);
    put_path(&pb, tree_t, logn, idx);
    for (int i = 0; i < PW; ++i) row[i] = lde_p[((size_t)i * 2 + c) * h + m];
    put(&pb, row, (size_t)PW);
    put_path(&pb, tree_p, logn, idx);
    for (int i = 0; i < 8; ++i) row[i] = lde_q[((size_t)i * 2 + c) * h + m];
    put(&pb, row, 8);
    put_path(&pb, tree_q, logn, idx);
    for (int k = 0; k < logh; ++k) {
      int loghk = logh - k;
      size_t hk = (size_t)1 << loghk, half = hk >> 1;
      size_t mk = m & (half - 1);
      size_t leaf = c * half + mk;
      put(&pb, fri_layer[k] + 4 * (c * hk + mk), 4);
      put(&pb, fri_layer[k] + 4 * (c * hk + mk + half), 4);
      put_path(&pb, fri_tree[k], loghk, leaf);
    }
  }
  free(row);
  for (int k = 0; k < logh; ++k) {
    free(fri_tree[k]);
    free(fri_layer[k]);
  }
  free(fri_tree);
  free(fri_layer);
  free(lde_t);
  free(coef_t);
  free(tree_t);
  free(lde_p);
  free(coef_p);
  free(tree_p);
  free(lde_q);
  free(coef_q);
  free(tree_q);
  if (pb.n * 4 != need) return 4;
  return 0;
}
