/* ORACLE -- TEST INFRASTRUCTURE ONLY (see field.h and machine.h).
 *
 * Whole machine proof on the CPU: the multi-table STARK that sp1-stark / sp1-prover 3.4.0 build over
 * p3-uni-stark, p3-fri, p3-merkle-tree (mixed-height MMCS) and p3-challenger (reference
 * Cargo.lock:7485, :7273, :5378, :5253, :5336, :5197) beneath `client.prove(&pk, stdin).run()`
 * (prover/src/bin/main.rs:71-74), restated under this repository's own format "ZKSP v12"
 * (DESIGN.md "Machine proof").  PARITY UNPINNED vs SP1 proof bytes.  The HIP prover must
 * reproduce these bytes exactly.
 *
 *   rounds      0 preprocessed (Program, Image; part of the verifying key), 1 main traces,
 *               2 LogUp permutation traces (helper columns + running sum), 3 quotient chunks
 *   commitment  one Poseidon2 Merkle tree per round over ALL chips: a leaf is the hash of the rows
 *               of the tallest matrices; a shorter matrix is injected one level up per halving,
 *               node = compress(compress(left, right), hash(rows)).  Tree position of LDE point
 *               (coset c, index m) of a height-H matrix: c * H + bitreverse(m), so that position >> d
 *               is the matching point (c, m mod H / 2^d) of a matrix 2^d times shorter.
 *   FRI         input of height 2^k joins the folding when the folded layer reaches 2^k points per
 *               coset (index-aligned addition), as in p3-fri's multi-height commit phase
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "machine.h"

static uint32_t bitrev32(uint32_t v, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}
static int ceil_log2(size_t v) {
  int l = 0;
  while (((size_t)1 << l) < v) ++l;
  return l;
}
static fe4 eval_poly(const uint32_t* coef, size_t n, fe4 z) {
  fe4 acc = e_zero();
  for (size_t k = n; k-- > 0;) acc = e_add(e_mul(acc, z), e_from(coef[k]));
  return acc;
}

typedef struct { uint32_t* w; size_t n, cap; } wbuf;
static void put(wbuf* b, const uint32_t* v, size_t n) {
  if (b->n + n <= b->cap) memcpy(b->w + b->n, v, n * 4);
  b->n += n;
}
static void observe_word_halves(orc_challenger* ch, const uint32_t* w, int n) {
  for (int i = 0; i < n; ++i) { orc_ch_observe(ch, w[i] & 0xffff); orc_ch_observe(ch, w[i] >> 16); }
}
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

/* ---- per-chip working set ---- */
enum { R_PREP = 0, R_MAIN, R_PERM, R_QUOT, N_ROUNDS };
typedef struct {
  const orc_chip* def;
  int logh;
  size_t h;
  int w[N_ROUNDS];
  uint32_t* tr[N_ROUNDS];    /* traces / quotient values [w][H] */
  uint32_t* lde[N_ROUNDS];   /* [w][2][H] */
  uint32_t* coef[N_ROUNDS];  /* [w][H] */
  fe4 cum;
} chipd;

/* ---- mixed-height commitment ---- */
typedef struct { int logn; uint32_t** level; } mmcs;

static void group_hash(const chipd* cd, int round, int group_logn, size_t pos, uint32_t* out8, uint32_t* rowbuf) {
  size_t n = 0;
  for (int c = 0; c < N_CHIPS; ++c) {
    const chipd* d = &cd[c];
    if (d->w[round] == 0 || d->logh + 1 != group_logn) continue;
    const size_t cs = pos >> d->logh, m = bitrev32((uint32_t)(pos & (d->h - 1)), d->logh);
    for (int col = 0; col < d->w[round]; ++col) rowbuf[n++] = d->lde[round][((size_t)col * 2 + cs) * d->h + m];
  }
  orc_hash_elems(rowbuf, n, out8);
}
static int group_exists(const chipd* cd, int round, int group_logn) {
  for (int c = 0; c < N_CHIPS; ++c)
    if (cd[c].w[round] && cd[c].logh + 1 == group_logn) return 1;
  return 0;
}
static int round_width(const chipd* cd, int round) {
  int w = 0;
  for (int c = 0; c < N_CHIPS; ++c) w += cd[c].w[round];
  return w;
}

static void mmcs_commit(const chipd* cd, int round, mmcs* t) {
  int lm = 0;
  for (int c = 0; c < N_CHIPS; ++c)
    if (cd[c].w[round] && cd[c].logh > lm) lm = cd[c].logh;
  t->logn = lm + 1;
  t->level = (uint32_t**)calloc((size_t)t->logn + 1, sizeof(uint32_t*));
  const int wmax = round_width(cd, round);
  for (int l = 0; l <= t->logn; ++l) {
    const size_t cnt = (size_t)1 << (t->logn - l);
    t->level[l] = (uint32_t*)malloc(8 * cnt * 4);
    const int inject = group_exists(cd, round, t->logn - l);
#pragma omp parallel
    {
      uint32_t* rowbuf = (uint32_t*)malloc((size_t)(wmax + 1) * 4);
#pragma omp for schedule(static)
      for (size_t p = 0; p < cnt; ++p) {
        uint32_t* d = t->level[l] + 8 * p;
        if (l == 0) {
          group_hash(cd, round, t->logn, p, d, rowbuf);
        } else {
          orc_compress(t->level[l - 1] + 16 * p, t->level[l - 1] + 16 * p + 8, d);
          if (inject) {
            uint32_t g[8], r[8];
            group_hash(cd, round, t->logn - l, p, g, rowbuf);
            orc_compress(d, g, r);
            memcpy(d, r, 32);
          }
        }
      }
      free(rowbuf);
    }
  }
}
static const uint32_t* mmcs_root(const mmcs* t) { return t->level[t->logn]; }
static void mmcs_free(mmcs* t) {
  for (int l = 0; l <= t->logn; ++l) free(t->level[l]);
  free(t->level);
}
/* rows of every chip in the round at the query point, then the path */
static void mmcs_open(const chipd* cd, int round, const mmcs* t, size_t cs, size_t m_max, wbuf* pb) {
  const int lm = t->logn - 1;
  for (int c = 0; c < N_CHIPS; ++c) {
    const chipd* d = &cd[c];
    if (!d->w[round]) continue;
    const size_t m = m_max & (d->h - 1);
    for (int col = 0; col < d->w[round]; ++col) put(pb, &d->lde[round][((size_t)col * 2 + cs) * d->h + m], 1);
  }
  const size_t hm = (size_t)1 << lm;
  const size_t pos = cs * hm + bitrev32((uint32_t)(m_max & (hm - 1)), lm);
  for (int l = 0; l < t->logn; ++l) put(pb, t->level[l] + 8 * ((pos >> l) ^ 1), 8);
}

/* ---- LogUp: fingerprints as sparse affine extension-field forms of the row ---- */
typedef struct { int n; int col[LF_MAX * INTER_MAX_ELEMS]; fe4 coef[LF_MAX * INTER_MAX_ELEMS]; fe4 c0; } aff;

static void aff_add(aff* a, int col, fe4 v) {
  for (int i = 0; i < a->n; ++i)
    if (a->col[i] == col) { a->coef[i] = e_add(a->coef[i], v); return; }
  a->col[a->n] = col;
  a->coef[a->n] = v;
  a->n++;
}
static void build_aff(const orc_inter* it, fe4 gamma, const fe4* bpow, aff* a) {
  a->n = 0;
  a->c0 = e_add(gamma, e_from((fe)it->bus));
  for (int j = 0; j < it->n_el; ++j) {
    const orc_lf* f = &it->el[j];
    a->c0 = e_add(a->c0, e_mul_base(bpow[j + 1], f->c0));
    for (int t = 0; t < f->n; ++t) aff_add(a, f->col[t], e_mul_base(bpow[j + 1], f->coef[t]));
  }
}
static inline fe4 aff_eval(const aff* a, const uint32_t* row) {
  fe4 f = a->c0;
  for (int i = 0; i < a->n; ++i) f = e_add(f, e_mul_base(a->coef[i], row[a->col[i]]));
  return f;
}
static inline fe lf_eval(const orc_lf* f, const uint32_t* row) {
  fe v = f->c0;
  for (int i = 0; i < f->n; ++i) v = f_add(v, f_mul(f->coef[i], row[f->col[i]]));
  return v;
}
/* extension-field versions for rows of extension elements are not needed by the prover */

static void gather_row(const chipd* d, const uint32_t* const* src, int is_lde, size_t cs, size_t m, uint32_t* row) {
  /* row = [prep | main]; src[R_PREP], src[R_MAIN] are traces ([w][H]) or LDEs ([w][2][H]) */
  int n = 0;
  for (int r = R_PREP; r <= R_MAIN; ++r)
    for (int col = 0; col < d->w[r]; ++col)
      row[n++] = is_lde ? src[r][((size_t)col * 2 + cs) * d->h + m] : src[r][(size_t)col * d->h + m];
}

/* An honest prover's own check: every base constraint of `chip` holds on every row of its trace (the rows taken
 * cyclically, with the selectors' values on the trace domain: is_first = [row 0], is_last = [last row], is_trans = 0 on the
 * last row only).  Returns the number of rows that violate one.  (A violated constraint does not show in the FRI layers:
 * with two quotient chunks over the two cosets of a blow-up of two, whatever was committed is of low degree - it shows in
 * the verifier's identity at zeta.) */
static size_t check_constraints(const chipd* d, int chip, const uint32_t* pub) {
  const int nb = d->def->n_constraints, rw = d->w[R_PREP] + d->w[R_MAIN], pw = d->w[R_PREP];
  const uint32_t* src[2] = {d->tr[R_PREP], d->tr[R_MAIN]};
  size_t bad = 0;
  if (nb == 0) return 0;
#pragma omp parallel reduction(+ : bad)
  {
    uint32_t* loc = (uint32_t*)malloc((size_t)rw * 4 * 2);
    uint32_t* nxt = loc + rw;
    uint32_t* cons = (uint32_t*)malloc((size_t)(nb + 1) * 4);
#pragma omp for schedule(static)
    for (size_t m = 0; m < d->h; ++m) {
      gather_row(d, src, 0, 0, m, loc);
      gather_row(d, src, 0, 0, (m + 1) & (d->h - 1), nxt);
      orc_machine_constraints(chip, loc, loc + pw, nxt + pw, m == 0, m == d->h - 1, m != d->h - 1, pub, cons);
      int any = 0;
      for (int k = 0; k < nb; ++k) any |= cons[k] != 0;
      bad += (size_t)any;
    }
    free(loc);
    free(cons);
  }
  return bad;
}

/* ---- LogUp slots (machine.h "LogUp layout") ---- */
static fe signed_mult(const orc_inter* it, const uint32_t* row) {
  const fe m = lf_eval(&it->mult, row);
  return it->sign < 0 ? f_neg(m) : m;
}
/* denominators and numerators of slot s: a regular pair / single gives (ma, fa, mb, fb) with (0, 1) for a missing second
 * fraction; the merged slot gives (M, F, 0, 1) */
static void slot_terms(const orc_chip* def, const aff* af, int s, const uint32_t* row, fe* ma, fe4* fa, fe* mb, fe4* fb) {
  const int nr = def->n_inter - def->n_merged, np = (nr + 1) / 2;
  *mb = 0;
  *fb = e_one();
  if (s < np) {
    *ma = signed_mult(&def->inter[2 * s], row);
    *fa = aff_eval(&af[2 * s], row);
    if (2 * s + 1 < nr) {
      *mb = signed_mult(&def->inter[2 * s + 1], row);
      *fb = aff_eval(&af[2 * s + 1], row);
    }
    return;
  }
  fe msum = 0;
  fe4 f = e_zero();
  for (int k = nr; k < def->n_inter; ++k) {
    const fe m = signed_mult(&def->inter[k], row);
    msum = f_add(msum, m);
    if (m) f = e_add(f, e_mul_base(aff_eval(&af[k], row), m));
  }
  *ma = msum;
  *fa = e_add(f, e_from(f_sub(1, msum)));
}
static fe4 slot_value(const orc_chip* def, const aff* af, int s, const uint32_t* row) {
  fe ma, mb;
  fe4 fa, fb, v = e_zero();
  slot_terms(def, af, s, row, &ma, &fa, &mb, &fb);
  if (ma) v = e_add(v, e_mul_base(e_inv(fa), ma));
  if (mb) v = e_add(v, e_mul_base(e_inv(fb), mb));
  return v;
}
/* v fa fb - (ma fb + mb fa): zero exactly when v is the slot's value */
static fe4 slot_constraint(const orc_chip* def, const aff* af, int s, const uint32_t* row, fe4 v) {
  fe ma, mb;
  fe4 fa, fb;
  slot_terms(def, af, s, row, &ma, &fa, &mb, &fb);
  return e_sub(e_mul(e_mul(v, fa), fb), e_add(e_mul_base(fb, ma), e_mul_base(fa, mb)));
}

static void perm_trace(chipd* d, fe4 gamma, const fe4* bpow) {
  const orc_chip* def = d->def;
  const int ni = def->n_inter, ns = orc_chip_slots(def), nh = ns - 1, rw = d->w[R_PREP] + d->w[R_MAIN];
  const size_t h = d->h;
  aff* af = (aff*)malloc((size_t)ni * sizeof(aff));
  for (int i = 0; i < ni; ++i) build_aff(&def->inter[i], gamma, bpow, &af[i]);
  uint32_t* p = d->tr[R_PERM];
  fe4* rowsum = (fe4*)malloc(h * sizeof(fe4));
  const uint32_t* src[2] = {d->tr[R_PREP], d->tr[R_MAIN]};
#pragma omp parallel
  {
    uint32_t* row = (uint32_t*)malloc((size_t)rw * 4);
#pragma omp for schedule(static)
    for (size_t r = 0; r < h; ++r) {
      gather_row(d, src, 0, 0, r, row);
      fe4 tot = e_zero();
      for (int j = 0; j < ns; ++j) {
        const fe4 hj = slot_value(def, af, j, row);
        if (j < nh)
          for (int t = 0; t < 4; ++t) p[(size_t)(4 * j + t) * h + r] = hj.c[t];
        tot = e_add(tot, hj);
      }
      rowsum[r] = tot;
    }
    free(row);
  }
  fe4 cum = e_zero();
  for (size_t r = 0; r < h; ++r) cum = e_add(cum, rowsum[r]);
  d->cum = cum;
  /* phi_0 = 0, phi_{r+1} = phi_r + rowsum_r - cum / H: back at 0 after the last row */
  const fe4 step = e_mul_base(cum, f_inv((fe)(h % FP)));
  fe4 acc = e_zero();
  for (size_t r = 0; r < h; ++r) {
    for (int t = 0; t < 4; ++t) p[(size_t)(4 * nh + t) * h + r] = acc.c[t];
    acc = e_sub(e_add(acc, rowsum[r]), step);
  }
  free(rowsum);
  free(af);
}

/* ---- quotient of one chip ---- */
/* adds the chip's constraints, folded with alpha^(alpha_off + k) and divided by the vanishing polynomial, to out [8][H] */
static void chip_quotient(chipd* d, int chip, fe4 alpha, int alpha_off, fe4 gamma, const fe4* bpow, const uint32_t* pub, uint32_t* out) {
  const orc_chip* def = d->def;
  const int ni = def->n_inter, nh = orc_chip_helpers(def), nb = def->n_constraints, total = nb + nh + 1;
  const fe4 cum_step = e_mul_base(d->cum, f_inv((fe)(d->h % FP)));
  const int rw = d->w[R_PREP] + d->w[R_MAIN], pw = d->w[R_PREP];
  const size_t h = d->h;
  fe4* apow = (fe4*)malloc((size_t)total * sizeof(fe4));
  apow[0] = e_one();
  for (int k = 0; k < alpha_off; ++k) apow[0] = e_mul(apow[0], alpha);
  for (int k = 1; k < total; ++k) apow[k] = e_mul(apow[k - 1], alpha);
  aff* af = (aff*)malloc((size_t)ni * sizeof(aff));
  for (int i = 0; i < ni; ++i) build_aff(&def->inter[i], gamma, bpow, &af[i]);
  const fe wh = f_root_of_unity(d->logh), w2h = f_root_of_unity(d->logh + 1), wh_inv = f_inv(wh);
  const uint32_t* src[2] = {d->lde[R_PREP], d->lde[R_MAIN]};
  const uint32_t* lp = d->lde[R_PERM];
  for (int cs = 0; cs < 2; ++cs) {
    const fe shift = cs ? f_mul(F_GEN, w2h) : F_GEN;
    const fe zh = f_sub(f_pow(shift, h), 1), zh_inv = f_inv(zh);
#pragma omp parallel
    {
      uint32_t* loc = (uint32_t*)malloc((size_t)rw * 4 * 2);
      uint32_t* nxt = loc + rw;
      uint32_t* cons = (uint32_t*)malloc((size_t)(nb + 1) * 4);
#pragma omp for schedule(static)
      for (size_t m = 0; m < h; ++m) {
        const size_t mn = (m + 1) & (h - 1);
        gather_row(d, src, 1, (size_t)cs, m, loc);
        gather_row(d, src, 1, (size_t)cs, mn, nxt);
        const fe x = f_mul(shift, f_pow(wh, m));
        const fe is_first = f_mul(zh, f_inv(f_sub(x, 1)));
        const fe is_last = f_mul(zh, f_inv(f_sub(x, wh_inv)));
        const fe is_trans = f_sub(x, wh_inv);
        orc_machine_constraints(chip, loc, loc + pw, nxt + pw, is_first, is_last, is_trans, pub, cons);
        fe4 acc = e_zero();
        for (int k = 0; k < nb; ++k) acc = e_add(acc, e_mul_base(apow[k], cons[k]));
        /* LogUp: the helper columns' slots, then the last slot, whose value is phi_next - phi + cum / H - sum of helpers */
        fe4 hsum = e_zero();
        for (int j = 0; j < nh; ++j) {
          fe4 hj;
          for (int t = 0; t < 4; ++t) hj.c[t] = lp[((size_t)(4 * j + t) * 2 + cs) * h + m];
          hsum = e_add(hsum, hj);
          acc = e_add(acc, e_mul(apow[nb + j], slot_constraint(def, af, j, loc, hj)));
        }
        fe4 phi, phin;
        for (int t = 0; t < 4; ++t) {
          phi.c[t] = lp[((size_t)(4 * nh + t) * 2 + cs) * h + m];
          phin.c[t] = lp[((size_t)(4 * nh + t) * 2 + cs) * h + mn];
        }
        const fe4 last = e_sub(e_add(e_sub(phin, phi), cum_step), hsum);
        acc = e_add(acc, e_mul(apow[nb + nh], slot_constraint(def, af, nh, loc, last)));
        acc = e_mul_base(acc, zh_inv);
        for (int t = 0; t < 4; ++t) out[(size_t)(4 * cs + t) * h + m] = f_add(out[(size_t)(4 * cs + t) * h + m], acc.c[t]);
      }
      free(loc);
      free(cons);
    }
  }
  free(apow);
  free(af);
}

static void lde_round(chipd* d, int round) {
  const size_t h = d->h;
  const int w = d->w[round];
  if (!w) return;
  d->lde[round] = (uint32_t*)malloc((size_t)w * 2 * h * 4);
  d->coef[round] = (uint32_t*)malloc((size_t)w * h * 4);
  if (round != R_QUOT) {
    orc_coset_lde(d->tr[round], d->logh, w, 1, d->lde[round], d->coef[round]);
  } else {
    const fe w2h = f_root_of_unity(d->logh + 1);
    for (int cs = 0; cs < 2; ++cs)
      orc_coset_lde(d->tr[round] + (size_t)4 * cs * h, d->logh, 4, cs ? f_mul(F_GEN, w2h) : F_GEN,
                    d->lde[round] + (size_t)4 * cs * 2 * h, d->coef[round] + (size_t)4 * cs * h);
  }
}

static void init_chips(const orc_machine_input* in, chipd* cd, int only_prep) {
  int logh[N_CHIPS];
  orc_machine_heights(in, logh);
  for (int c = 0; c < N_CHIPS; ++c) {
    chipd* d = &cd[c];
    memset(d, 0, sizeof *d);
    d->def = orc_machine_chip(c);
    d->logh = logh[c];
    d->h = (size_t)1 << logh[c];
    d->w[R_PREP] = d->def->prep_width;
    if (only_prep && !d->w[R_PREP]) continue;
    d->w[R_MAIN] = d->def->main_width;
    if (d->w[R_PREP]) d->tr[R_PREP] = (uint32_t*)malloc((size_t)d->w[R_PREP] * d->h * 4);
    d->tr[R_MAIN] = (uint32_t*)malloc((size_t)d->w[R_MAIN] * d->h * 4);
    orc_machine_fill(in, c, d->logh, d->tr[R_PREP], d->tr[R_MAIN]);
    lde_round(d, R_PREP);
    if (only_prep) { d->w[R_MAIN] = 0; continue; }
    d->w[R_PERM] = orc_chip_perm_width(d->def);
    d->w[R_QUOT] = orc_quot_leader(logh, c) == c ? 8 : 0; /* one quotient per height */
  }
}
static void free_chips(chipd* cd) {
  for (int c = 0; c < N_CHIPS; ++c)
    for (int r = 0; r < N_ROUNDS; ++r) { free(cd[c].tr[r]); free(cd[c].lde[r]); free(cd[c].coef[r]); }
}

/* ---- kernel-level parity: one chip's LogUp permutation trace and quotient values for given challenges ---- */
static void one_chip(const orc_machine_input* in, int chip, chipd* d) {
  int logh[N_CHIPS];
  orc_machine_heights(in, logh);
  memset(d, 0, sizeof *d);
  d->def = orc_machine_chip(chip);
  d->logh = logh[chip];
  d->h = (size_t)1 << logh[chip];
  d->w[R_PREP] = d->def->prep_width; d->w[R_MAIN] = d->def->main_width; d->w[R_PERM] = orc_chip_perm_width(d->def); d->w[R_QUOT] = 8;
  if (d->w[R_PREP]) d->tr[R_PREP] = (uint32_t*)malloc((size_t)d->w[R_PREP] * d->h * 4);
  d->tr[R_MAIN] = (uint32_t*)malloc((size_t)d->w[R_MAIN] * d->h * 4);
  orc_machine_fill(in, chip, d->logh, d->tr[R_PREP], d->tr[R_MAIN]);
}
static void free_one(chipd* d) {
  for (int r = 0; r < N_ROUNDS; ++r) { free(d->tr[r]); free(d->lde[r]); free(d->coef[r]); }
}
static void challenge_powers(const uint32_t gamma4[4], const uint32_t beta4[4], fe4* gamma, fe4* bpow) {
  fe4 beta;
  memcpy(gamma->c, gamma4, 16);
  memcpy(beta.c, beta4, 16);
  bpow[0] = e_one();
  for (int j = 1; j <= INTER_MAX_ELEMS; ++j) bpow[j] = e_mul(bpow[j - 1], beta);
}
void orc_machine_stage_perm(const orc_machine_input* in, int chip, const uint32_t gamma4[4], const uint32_t beta4[4], uint32_t* perm,
                            uint32_t cum[4]) {
  chipd d;
  fe4 gamma, bpow[INTER_MAX_ELEMS + 1];
  one_chip(in, chip, &d);
  challenge_powers(gamma4, beta4, &gamma, bpow);
  d.tr[R_PERM] = (uint32_t*)calloc((size_t)d.w[R_PERM] * d.h, 4);
  perm_trace(&d, gamma, bpow);
  memcpy(perm, d.tr[R_PERM], (size_t)d.w[R_PERM] * d.h * 4);
  memcpy(cum, d.cum.c, 16);
  free_one(&d);
}
void orc_machine_stage_quotient(const orc_machine_input* in, int chip, const uint32_t alpha4[4], const uint32_t gamma4[4],
                                const uint32_t beta4[4], uint32_t* quot) {
  chipd d;
  fe4 gamma, alpha, bpow[INTER_MAX_ELEMS + 1];
  uint32_t pub[CPUPUB_N] = {0, 0, 0, 0, 0};
  one_chip(in, chip, &d);
  challenge_powers(gamma4, beta4, &gamma, bpow);
  memcpy(alpha.c, alpha4, 16);
  if (orc_cpu_instance(chip) >= 0) orc_machine_cpu_pub(in, chip, pub);
  if (chip == CH_ECALL) orc_machine_cpu_pub(in, CH_CPU, pub); /* (the padding pc) */
  lde_round(&d, R_PREP);
  lde_round(&d, R_MAIN);
  d.tr[R_PERM] = (uint32_t*)calloc((size_t)d.w[R_PERM] * d.h, 4);
  perm_trace(&d, gamma, bpow);
  lde_round(&d, R_PERM);
  int logh[N_CHIPS];
  orc_machine_heights(in, logh);
  d.tr[R_QUOT] = (uint32_t*)calloc((size_t)8 * d.h, 4);
  chip_quotient(&d, chip, alpha, orc_quot_alpha_offset(logh, chip), gamma, bpow, pub, d.tr[R_QUOT]);
  memcpy(quot, d.tr[R_QUOT], (size_t)8 * d.h * 4);
  free_one(&d);
}

static void vk_digest_of(const uint32_t root[8], const orc_machine_input* in, int keccak_mode, uint32_t out[8]) {
  uint32_t v[18];
  const uint32_t pad_pc = in->text_base + 4 * (uint32_t)(in->n_program - 1); /* the padding instruction: last Program row */
  memcpy(v, root, 32);
  v[8] = in->entry & 0xffff; v[9] = in->entry >> 16;
  v[10] = (uint32_t)in->log_prog; v[11] = (uint32_t)in->log_image; v[12] = (uint32_t)keccak_mode;
  v[13] = ZKSP_VERSION_MACHINE; v[14] = CPU_WIDTH; v[15] = N_CHIPS;
  v[16] = pad_pc & 0xffff; v[17] = pad_pc >> 16;
  orc_hash_elems(v, 18, out);
}

void orc_machine_setup(const orc_machine_input* in, int keccak_mode, uint32_t prep_root[8], uint32_t vk_digest[8]) {
  chipd cd[N_CHIPS];
  orc_machine_input tmp = *in;
  tmp.prog_mult = NULL;
  tmp.shape = NULL;
  tmp.agg_leaves = NULL;
  tmp.agg_keys = NULL;
  tmp.n_agg = 0;
  tmp.leaf_p2_rows = tmp.leaf_qr_rows = tmp.leaf_tr_rows = tmp.pub_tuples = NULL;
  tmp.n_leaf_p2 = tmp.n_leaf_qr = tmp.n_leaf_tr = tmp.n_pub = 0;
  tmp.n_cycles = tmp.n_keccak = tmp.n_memfinal = tmp.n_muls = 0;
  init_chips(&tmp, cd, 1);
  mmcs t;
  mmcs_commit(cd, R_PREP, &t);
  memcpy(prep_root, mmcs_root(&t), 32);
  vk_digest_of(prep_root, in, keccak_mode, vk_digest);
  mmcs_free(&t);
  free_chips(cd);
}

/* magic, version, heights, exit code, pv length, 3 digests, hand-over pc; aggregation: leaf count, root, digest of the leaf list;
 * public bus tuples (the statement of a leaf-proof check): count, digest of the list */
#define HEADER_WORDS (2 + N_CHIPS + 2 + 24 + (CPU_INST - 1) + 17 + 9)

typedef struct { uint32_t key; int have; uint32_t d[8]; } agg_node;
static int agg_node_cmp(const void* x, const void* y) {
  const uint32_t a = ((const agg_node*)x)->key, b = ((const agg_node*)y)->key;
  return a < b ? -1 : a > b;
}
static agg_node* agg_find(agg_node* v, size_t n, uint32_t key) {
  agg_node probe;
  probe.key = key;
  return (agg_node*)bsearch(&probe, v, n, sizeof(agg_node), agg_node_cmp);
}
size_t orc_machine_agg_rows(const uint32_t* keys, const uint32_t* digests, size_t n, uint32_t* rows) {
  if (n == 0) return 0;
  /* supplied nodes and their ancestors, one entry per key */
  size_t cap = n * 32 + 1, cnt = 0;
  agg_node* v = (agg_node*)malloc(cap * sizeof(agg_node));
  for (size_t j = 0; j < n; ++j) {
    const uint32_t key = keys ? keys[j] : (uint32_t)(n + j);
    if (key < 2 || key >= (1u << 30)) { free(v); return (size_t)-1; }
    v[cnt].key = key; v[cnt].have = 1; memcpy(v[cnt].d, digests + 8 * j, 32); ++cnt;
  }
  qsort(v, cnt, sizeof(agg_node), agg_node_cmp);
  for (size_t j = 1; j < cnt; ++j) if (v[j].key == v[j - 1].key) { free(v); return (size_t)-1; }
  const size_t n_sup = cnt;
  for (size_t j = 0; j < n_sup; ++j)
    for (uint32_t k = v[j].key >> 1; k >= 1; k >>= 1) {
      /* (ancestors are appended unsorted, duplicates and all, then sorted and deduplicated) */
      if (cnt == cap) { cap *= 2; v = (agg_node*)realloc(v, cap * sizeof(agg_node)); }
      v[cnt].key = k; v[cnt].have = 0; ++cnt;
      if (k == 1) break;
    }
  /* supplied entries first among equal keys: a supplied key that is also an ancestor is malformed */
  qsort(v, cnt, sizeof(agg_node), agg_node_cmp);
  size_t m = 0;
  for (size_t j = 0; j < cnt; ++j) {
    if (m && v[m - 1].key == v[j].key) {
      if (v[j].have || v[m - 1].have) { free(v); return (size_t)-1; }
      continue;
    }
    v[m++] = v[j];
  }
  cnt = m;
  size_t n_rows = 0;
  for (size_t j = cnt; j-- > 0;) { /* descending keys: children before parents */
    if (v[j].have) continue;
    agg_node *l = agg_find(v, cnt, 2 * v[j].key), *r = agg_find(v, cnt, 2 * v[j].key + 1);
    if (!l || !r || !l->have || !r->have) { free(v); return (size_t)-1; }
    orc_compress(l->d, r->d, v[j].d);
    v[j].have = 2; /* computed */
    ++n_rows;
  }
  if (rows) {
    size_t r = 0;
    for (size_t j = 0; j < cnt; ++j) { /* ascending keys: the root first */
      if (v[j].have != 2) continue;
      uint32_t* o = rows + 25 * r++;
      o[0] = v[j].key;
      memcpy(o + 1, agg_find(v, cnt, 2 * v[j].key)->d, 32);
      memcpy(o + 9, agg_find(v, cnt, 2 * v[j].key + 1)->d, 32);
      memcpy(o + 17, v[j].d, 32);
    }
  }
  free(v);
  return n_rows;
}
int orc_machine_nodes_public(const uint32_t* keys, const uint32_t* digests, size_t n, uint32_t root[8], uint32_t list_digest[8]) {
  memset(root, 0, 32);
  memset(list_digest, 0, 32);
  if (n == 0) return 1;
  const size_t n_rows = orc_machine_agg_rows(keys, digests, n, NULL);
  if (n_rows == (size_t)-1 || n_rows == 0) return 0;
  uint32_t* rows = (uint32_t*)malloc(n_rows * 25 * 4);
  orc_machine_agg_rows(keys, digests, n, rows);
  memcpy(root, rows + 17, 32); /* row 0 is key 1 */
  const int ok = rows[0] == 1;
  free(rows);
  /* the list in the transcript: every key with its digest */
  uint32_t* flat = (uint32_t*)malloc(n * 9 * 4);
  for (size_t j = 0; j < n; ++j) {
    flat[9 * j] = keys ? keys[j] : (uint32_t)(n + j);
    memcpy(flat + 9 * j + 1, digests + 8 * j, 32);
  }
  orc_hash_elems(flat, 9 * n, list_digest);
  free(flat);
  return ok;
}
void orc_machine_agg_public(const uint32_t* leaves, size_t n, uint32_t root[8], uint32_t list_digest[8]) {
  if (!orc_machine_nodes_public(NULL, leaves, n, root, list_digest)) { memset(root, 0, 32); memset(list_digest, 0, 32); }
}

/* Coefficients of the reduced openings (format v16).  The input of height 2^lh at the LDE point x is
 *   sum over rounds r of delta^r (H_r(x) - H_r(zeta)) / (x - zeta)  +  sum over r = main, permutation of
 *   delta^(3 + r) (H_r(x) - H_r(zeta w)) / (x - zeta w),
 * where H_r is Horner's rule in alpha_f over the SEGMENT (r, lh): the opened rows of the chips of that height in chip order,
 * zero-filled to a multiple of eight words - exactly the words the opening's sponge absorbs, in the order it absorbs them,
 * so that a proof ABOUT the opening can accumulate H_r block by block (machine.h, Poseidon2 chip).  Word i of a segment of
 * n words has the coefficient delta^r alpha_f^(8 ceil(n / 8) - 1 - i).  out[open index]: as the opened values are laid
 * out (per chip: prep, main, perm, quot at zeta, then main, perm at zeta w). */
void orc_reduce_coefs(const int logh[N_CHIPS], fe4 af, fe4 delta, fe4* out) {
  int wr[N_CHIPS][N_ROUNDS];
  size_t off[N_CHIPS], o = 0;
  int emax = 0;
  for (int c = 0; c < N_CHIPS; ++c) {
    const orc_chip* d = orc_machine_chip(c);
    wr[c][R_PREP] = d->prep_width; wr[c][R_MAIN] = d->main_width; wr[c][R_PERM] = orc_chip_perm_width(d);
    wr[c][R_QUOT] = orc_quot_leader(logh, c) == c ? 8 : 0;
    off[c] = o;
    o += (size_t)wr[c][R_PREP] + 2 * (size_t)wr[c][R_MAIN] + 2 * (size_t)wr[c][R_PERM] + (size_t)wr[c][R_QUOT];
  }
  int seg_len[32][N_ROUNDS];
  memset(seg_len, 0, sizeof seg_len);
  for (int c = 0; c < N_CHIPS; ++c)
    for (int r = 0; r < N_ROUNDS; ++r) seg_len[logh[c]][r] += wr[c][r];
  for (int l = 0; l < 32; ++l)
    for (int r = 0; r < N_ROUNDS; ++r)
      if (seg_len[l][r] > emax) emax = seg_len[l][r];
  emax += 8;
  fe4* ap = (fe4*)malloc((size_t)emax * sizeof(fe4));
  ap[0] = e_one();
  for (int i = 1; i < emax; ++i) ap[i] = e_mul(ap[i - 1], af);
  fe4 dp[6];
  dp[0] = e_one();
  for (int i = 1; i < 6; ++i) dp[i] = e_mul(dp[i - 1], delta);
  int pos[32][N_ROUNDS];
  memset(pos, 0, sizeof pos);
  for (int c = 0; c < N_CHIPS; ++c) {
    const int lh = logh[c];
    size_t i = 0, j = 0;
    const size_t n1 = (size_t)wr[c][R_PREP] + wr[c][R_MAIN] + wr[c][R_PERM] + wr[c][R_QUOT];
    for (int r = 0; r < N_ROUNDS; ++r) {
      const int lpad = (seg_len[lh][r] + 7) / 8 * 8;
      for (int col = 0; col < wr[c][r]; ++col, ++i) {
        const fe4 a = ap[lpad - 1 - (pos[lh][r] + col)];
        out[off[c] + i] = e_mul(dp[r], a);
        if (r == R_MAIN || r == R_PERM) { out[off[c] + n1 + j] = e_mul(dp[3 + r], a); ++j; }
      }
      pos[lh][r] += wr[c][r];
    }
  }
  free(ap);
}

size_t orc_machine_proof_size(const int logh[N_CHIPS], int log_prog, int log_image, const orc_config* cfg, uint32_t pv_len) {
  (void)log_prog; (void)log_image;
  size_t words = HEADER_WORDS + (pv_len + 3) / 4;
  int lm = 0, lm_prep = 0;
  size_t opened = 0, rw[N_ROUNDS] = {0, 0, 0, 0};
  for (int c = 0; c < N_CHIPS; ++c) {
    const orc_chip* d = orc_machine_chip(c);
    const size_t e = (size_t)orc_chip_perm_width(d);
    if (logh[c] > lm) lm = logh[c];
    if (d->prep_width && logh[c] > lm_prep) lm_prep = logh[c];
    const size_t q = orc_quot_leader(logh, c) == c ? 8 : 0;
    opened += (size_t)d->prep_width + 2 * (size_t)d->main_width + 2 * e + q;
    rw[R_PREP] += (size_t)d->prep_width; rw[R_MAIN] += (size_t)d->main_width; rw[R_PERM] += e; rw[R_QUOT] += q;
  }
  words += 8 + 8 + 4 * N_CHIPS + 8 + 4 * opened + 8 * (size_t)lm + 4 + 1;
  size_t perq = rw[R_PREP] + 8 * ((size_t)lm_prep + 1);
  for (int r = R_MAIN; r <= R_QUOT; ++r) perq += rw[r] + 8 * ((size_t)lm + 1);
  for (int k = 0; k < lm; ++k) perq += 8 + 8 * (size_t)(lm - k);
  words += perq * cfg->num_queries;
  return words * 4;
}

int orc_machine_prove(const orc_machine_input* in, int keccak_mode, const orc_machine_public* pub, const uint8_t* public_values,
                      const orc_config* cfg, uint8_t* out, size_t cap, size_t* out_len) {
  chipd cd[N_CHIPS];
  int logh[N_CHIPS];
  orc_machine_heights(in, logh);
  if (in->n_cycles < 1 || logh[CH_CPU] > 20) return 1; /* CPU instances of at most 2^20 rows each */
  const size_t need = orc_machine_proof_size(logh, in->log_prog, in->log_image, cfg, pub->pv_len);
  *out_len = need;
  if (cap < need) return 2;
  wbuf pb = {(uint32_t*)out, 0, cap / 4};
  int lm = 0; /* the tallest chip: every tree and the FRI start from its height */
  for (int c = 0; c < N_CHIPS; ++c)
    if (logh[c] > lm) lm = logh[c];
  uint32_t agg_n = (uint32_t)in->n_agg, agg_root[8], agg_digest[8];
  if (!orc_machine_nodes_public(in->agg_keys, in->agg_leaves, in->n_agg, agg_root, agg_digest)) return 1; /* malformed payload */
  /* the statement of a leaf-proof check: the list of public bus tuples stands in the header and the transcript by its digest */
  uint32_t pub_n = (uint32_t)in->n_pub, pub_digest[8];
  memset(pub_digest, 0, sizeof pub_digest);
  if (in->n_pub) {
    for (size_t i = 0; i < in->n_pub * PUB_TUPLE_WORDS; ++i)
      if (in->pub_tuples[i] >= FP) return 1;
    orc_hash_elems(in->pub_tuples, in->n_pub * PUB_TUPLE_WORDS, pub_digest);
  }
  uint32_t cpu_pub[N_CHIPS][CPUPUB_N];
  memset(cpu_pub, 0, sizeof cpu_pub);
  uint32_t handover[CPU_INST - 1]; /* the pc every later CPU instance starts at: header words, absorbed into the transcript */
  for (int i = 0; i < CPU_INST; ++i) {
    orc_machine_cpu_pub(in, orc_cpu_chip(i), cpu_pub[orc_cpu_chip(i)]);
    if (i + 1 < CPU_INST) handover[i] = cpu_pub[orc_cpu_chip(i)][CPUPUB_END_PC];
  }
  orc_machine_cpu_pub(in, CH_CPU, cpu_pub[CH_ECALL]); /* the ecall chip's constraints use the padding pc */

  /* ---- rounds 0 and 1: preprocessed and main traces ---- */
  init_chips(in, cd, 0);
  mmcs t_prep, t_main, t_perm, t_quot;
  mmcs_commit(cd, R_PREP, &t_prep);
  uint32_t vk[8];
  vk_digest_of(mmcs_root(&t_prep), in, keccak_mode, vk);
  /* the records are consistent only if every chip's constraints hold on its trace.  ZKSP_ORACLE_FORCE lets soundness
   * tests obtain the proof a cheating prover would send, to check that the verifier rejects it */
  for (int c = 0; c < N_CHIPS; ++c) {
    const size_t bad = check_constraints(&cd[c], c, cpu_pub[c]);
    if (bad && getenv("ZKSP_ORACLE_TIMING")) fprintf(stderr, "[oracle] chip %-10s: %zu rows violate a constraint\n", cd[c].def->name, bad);
    if (bad && !getenv("ZKSP_ORACLE_FORCE")) { free_chips(cd); return 3; }
  }
  for (int c = 0; c < N_CHIPS; ++c) lde_round(&cd[c], R_MAIN);
  mmcs_commit(cd, R_MAIN, &t_main);

  /* header */
  {
    uint32_t head[2] = {ZKSP_MAGIC, ZKSP_VERSION_MACHINE};
    put(&pb, head, 2);
    for (int c = 0; c < N_CHIPS; ++c) { uint32_t v = (uint32_t)logh[c]; put(&pb, &v, 1); }
    put(&pb, &pub->exit_code, 1);
    put(&pb, &pub->pv_len, 1);
    put(&pb, pub->pv_digest, 8);
    put(&pb, pub->deferred_digest, 8);
    put(&pb, vk, 8);
    put(&pb, handover, CPU_INST - 1);
    put(&pb, &agg_n, 1);
    put(&pb, agg_root, 8);
    put(&pb, agg_digest, 8);
    put(&pb, &pub_n, 1);
    put(&pb, pub_digest, 8);
    size_t pw = (pub->pv_len + 3) / 4;
    uint32_t* tmp = (uint32_t*)calloc(pw ? pw : 1, 4);
    memcpy(tmp, public_values, pub->pv_len);
    put(&pb, tmp, pw);
    free(tmp);
  }
  orc_challenger ch;
  orc_ch_init(&ch);
  orc_ch_observe_many(&ch, vk, 8);
  for (int c = 0; c < N_CHIPS; ++c) orc_ch_observe(&ch, (uint32_t)logh[c]);
  orc_ch_observe(&ch, pub->exit_code & 0xffff);
  orc_ch_observe(&ch, pub->exit_code >> 16);
  observe_word_halves(&ch, pub->pv_digest, 8);
  observe_word_halves(&ch, pub->deferred_digest, 8);
  observe_word_halves(&ch, handover, CPU_INST - 1);
  orc_ch_observe(&ch, agg_n);
  orc_ch_observe_many(&ch, agg_root, 8);
  orc_ch_observe_many(&ch, agg_digest, 8);
  orc_ch_observe(&ch, pub_n);
  orc_ch_observe_many(&ch, pub_digest, 8);
  orc_ch_pad(&ch); /* (v16: a commitment root is a block of its own) */
  orc_ch_observe_many(&ch, mmcs_root(&t_main), 8);
  orc_ch_pad(&ch); /* (v16: a phase ends on a block boundary) */
  put(&pb, mmcs_root(&t_main), 8);

  /* ---- round 2: LogUp ---- */
  fe4 gamma, beta, bpow[INTER_MAX_ELEMS + 1];
  orc_ch_sample_ext(&ch, gamma.c);
  orc_ch_sample_ext(&ch, beta.c);
  bpow[0] = e_one();
  for (int j = 1; j <= INTER_MAX_ELEMS; ++j) bpow[j] = e_mul(bpow[j - 1], beta);
  fe4 total = e_zero();
  for (int c = 0; c < N_CHIPS; ++c) {
    chipd* d = &cd[c];
    d->tr[R_PERM] = (uint32_t*)calloc((size_t)d->w[R_PERM] * d->h, 4);
    perm_trace(d, gamma, bpow);
    total = e_add(total, d->cum);
    lde_round(d, R_PERM);
  }
  /* the verifier closes the two public buses: digest words and exit code */
  {
    aff a;
    orc_inter it;
    memset(&it, 0, sizeof it);
    for (int kind = 1; kind <= 2; ++kind)
      for (int i = 0; i < 8; ++i) {
        const uint32_t w = kind == 1 ? pub->pv_digest[i] : pub->deferred_digest[i];
        it.bus = BUS_PUBC; it.n_el = 4;
        it.el[0].n = 0; it.el[0].c0 = (uint32_t)kind;
        it.el[1].n = 0; it.el[1].c0 = (uint32_t)i;
        it.el[2].n = 0; it.el[2].c0 = w & 0xffff;
        it.el[3].n = 0; it.el[3].c0 = w >> 16;
        build_aff(&it, gamma, bpow, &a);
        total = e_sub(total, e_inv(a.c0));
      }
    it.bus = BUS_PUBH; it.n_el = 2;
    it.el[0].n = 0; it.el[0].c0 = pub->exit_code & 0xffff;
    it.el[1].n = 0; it.el[1].c0 = pub->exit_code >> 16;
    build_aff(&it, gamma, bpow, &a);
    total = e_sub(total, e_inv(a.c0));
    /* ... and the digest bus of the aggregation payload: the verifier hands in the supplied nodes (the leaves of a full tree
     * at heap nodes n .. 2n - 1, or a leaf and the siblings along its path) and takes the root (node 1) */
    it.bus = BUS_DIGEST; it.n_el = 12; /* (tag 0, type 2 = heap node, key, mask 0, the digest) */
    for (size_t i = 0; i <= in->n_agg && in->n_agg; ++i) {
      const uint32_t* d = i < in->n_agg ? in->agg_leaves + 8 * i : agg_root;
      it.el[0].n = 0; it.el[0].c0 = 0;
      it.el[1].n = 0; it.el[1].c0 = 2;
      it.el[2].n = 0; it.el[2].c0 = i < in->n_agg ? (in->agg_keys ? in->agg_keys[i] : (uint32_t)(in->n_agg + i)) : 1u;
      it.el[3].n = 0; it.el[3].c0 = 0;
      for (int j = 0; j < 8; ++j) { it.el[4 + j].n = 0; it.el[4 + j].c0 = d[j]; }
      build_aff(&it, gamma, bpow, &a);
      total = i < in->n_agg ? e_add(total, e_inv(a.c0)) : e_sub(total, e_inv(a.c0));
    }
    /* ... and the public bus tuples of a leaf-proof check, each sent or received by the verifier with its multiplicity */
    for (size_t i = 0; i < in->n_pub; ++i) {
      const uint32_t* t = in->pub_tuples + PUB_TUPLE_WORDS * i;
      if (t[3] > PUB_TUPLE_WORDS - 4) return 1;
      it.bus = (int)t[0]; it.n_el = (int)t[3];
      for (uint32_t j = 0; j < t[3]; ++j) { it.el[j].n = 0; it.el[j].c0 = t[4 + j]; }
      build_aff(&it, gamma, bpow, &a);
      const fe4 term = e_mul_base(e_inv(a.c0), t[2]);
      total = t[1] ? e_add(total, term) : e_sub(total, term);
    }
  }
  if (!e_eq(total, e_zero())) {
    if (getenv("ZKSP_ORACLE_TIMING")) {
      for (int c = 0; c < N_CHIPS; ++c)
        fprintf(stderr, "[oracle] chip %-10s cumulative sum %u %u %u %u\n", cd[c].def->name, cd[c].cum.c[0], cd[c].cum.c[1],
                cd[c].cum.c[2], cd[c].cum.c[3]);
    }
    /* the buses do not balance: the records are inconsistent.  ZKSP_ORACLE_FORCE lets soundness tests
     * obtain the proof a cheating prover would send, to check that the verifier rejects it */
    if (!getenv("ZKSP_ORACLE_FORCE")) return 5;
  }
  mmcs_commit(cd, R_PERM, &t_perm);
  orc_ch_observe_many(&ch, mmcs_root(&t_perm), 8);
  put(&pb, mmcs_root(&t_perm), 8);
  for (int c = 0; c < N_CHIPS; ++c) {
    orc_ch_observe_many(&ch, cd[c].cum.c, 4);
    put(&pb, cd[c].cum.c, 4);
  }

  /* ---- round 3: quotients ---- */
  fe4 alpha;
  orc_ch_pad(&ch);
  orc_ch_sample_ext(&ch, alpha.c);
  for (int c = 0; c < N_CHIPS; ++c) { /* every chip adds its share to the quotient of its height's first chip */
    chipd* d = &cd[c];
    if (d->w[R_QUOT]) d->tr[R_QUOT] = (uint32_t*)calloc((size_t)8 * d->h, 4);
    chip_quotient(d, c, alpha, orc_quot_alpha_offset(logh, c), gamma, bpow, cpu_pub[c], cd[orc_quot_leader(logh, c)].tr[R_QUOT]);
  }
  for (int c = 0; c < N_CHIPS; ++c) lde_round(&cd[c], R_QUOT);
  mmcs_commit(cd, R_QUOT, &t_quot);
  orc_ch_observe_many(&ch, mmcs_root(&t_quot), 8);
  put(&pb, mmcs_root(&t_quot), 8);

  /* ---- openings at zeta (everything) and zeta * w_H (main, permutation) ---- */
  fe4 zeta;
  orc_ch_sample_ext(&ch, zeta.c);
  size_t n_open = 0;
  for (int c = 0; c < N_CHIPS; ++c) n_open += (size_t)cd[c].w[R_PREP] + 2 * (size_t)cd[c].w[R_MAIN] + 2 * (size_t)cd[c].w[R_PERM] + (size_t)cd[c].w[R_QUOT];
  fe4* opened = (fe4*)malloc(n_open * sizeof(fe4));
  size_t chip_open_off[N_CHIPS];
  {
    size_t o = 0;
    for (int c = 0; c < N_CHIPS; ++c) {
      chipd* d = &cd[c];
      chip_open_off[c] = o;
      const fe4 zn = e_mul_base(zeta, f_root_of_unity(d->logh));
      for (int pass = 0; pass < 2; ++pass)
        for (int r = (pass ? R_MAIN : R_PREP); r <= (pass ? R_PERM : R_QUOT); ++r) {
          const int w = d->w[r];
          fe4* dst = opened + o;
#pragma omp parallel for schedule(static)
          for (int i = 0; i < w; ++i) dst[i] = eval_poly(d->coef[r] + (size_t)i * d->h, d->h, pass ? zn : zeta);
          o += (size_t)w;
        }
    }
  }
  put(&pb, (const uint32_t*)opened, n_open * 4);
  observe_list_root(&ch, (const uint32_t*)opened, n_open * 4, ceil_log2((n_open * 4 + 7) / 8));

  /* ---- reduced openings, one input per height (v16: orc_reduce_coefs) ---- */
  fe4 af, delta;
  orc_ch_sample_ext(&ch, af.c);
  orc_ch_sample_ext(&ch, delta.c);
  fe4* afpow = (fe4*)malloc(n_open * sizeof(fe4));
  orc_reduce_coefs(logh, af, delta, afpow);
  fe4* G[32];
  memset(G, 0, sizeof G);
  for (int c = 0; c < N_CHIPS; ++c) {
    chipd* d = &cd[c];
    const size_t h = d->h, o = chip_open_off[c];
    const size_t n1 = (size_t)d->w[R_PREP] + d->w[R_MAIN] + d->w[R_PERM] + d->w[R_QUOT], n2 = (size_t)d->w[R_MAIN] + d->w[R_PERM];
    fe4 b1 = e_zero(), b2 = e_zero();
    for (size_t i = 0; i < n1; ++i) b1 = e_add(b1, e_mul(afpow[o + i], opened[o + i]));
    for (size_t i = 0; i < n2; ++i) b2 = e_add(b2, e_mul(afpow[o + n1 + i], opened[o + n1 + i]));
    if (!G[d->logh]) G[d->logh] = (fe4*)calloc(2 * h, sizeof(fe4));
    fe4* g = G[d->logh];
    const fe wh = f_root_of_unity(d->logh), w2h = f_root_of_unity(d->logh + 1);
    const fe4 zn = e_mul_base(zeta, wh);
    for (int cs = 0; cs < 2; ++cs) {
      const fe shift = cs ? f_mul(F_GEN, w2h) : F_GEN;
#pragma omp parallel for schedule(static)
      for (size_t m = 0; m < h; ++m) {
        const fe x = f_mul(shift, f_pow(wh, m));
        fe4 s1 = e_zero(), s2 = e_zero();
        size_t i = 0, j = 0;
        for (int r = R_PREP; r <= R_QUOT; ++r)
          for (int col = 0; col < d->w[r]; ++col, ++i) {
            const fe v = d->lde[r][((size_t)col * 2 + cs) * h + m];
            s1 = e_add(s1, e_mul_base(afpow[o + i], v));
            if (r == R_MAIN || r == R_PERM) { s2 = e_add(s2, e_mul_base(afpow[o + n1 + j], v)); ++j; }
          }
        const fe4 d0 = e_inv(e_sub(e_from(x), zeta)), d1 = e_inv(e_sub(e_from(x), zn));
        fe4 v = e_add(e_mul(e_sub(s1, b1), d0), e_mul(e_sub(s2, b2), d1));
        g[(size_t)cs * h + m] = e_add(g[(size_t)cs * h + m], v);
      }
    }
  }
  free(afpow);
  free(opened);

  /* ---- FRI commit phase with inputs joining at their height ---- */
  uint32_t** fri_tree = (uint32_t**)malloc((size_t)lm * sizeof(uint32_t*));
  uint32_t** fri_layer = (uint32_t**)malloc((size_t)lm * sizeof(uint32_t*));
  uint32_t* layer = (uint32_t*)malloc(((size_t)2 << lm) * 16);
  memcpy(layer, G[lm], ((size_t)2 << lm) * 16);
  fe shift_k = F_GEN;
  for (int k = 0; k < lm; ++k) {
    const int loghk = lm - k;
    const size_t hk = (size_t)1 << loghk, half = hk >> 1;
    uint32_t* mat = (uint32_t*)malloc(8 * hk * 4);
    for (int cs = 0; cs < 2; ++cs)
      for (size_t m = 0; m < half; ++m)
        for (int j = 0; j < 4; ++j) {
          mat[(size_t)j * hk + cs * half + m] = layer[4 * ((size_t)cs * hk + m) + j];
          mat[(size_t)(4 + j) * hk + cs * half + m] = layer[4 * ((size_t)cs * hk + m + half) + j];
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
    if (loghk - 1 >= 0 && G[loghk - 1]) {
      const fe4* g = G[loghk - 1];
      for (size_t i = 0; i < hk; ++i) {
        fe4 v;
        memcpy(v.c, nxt + 4 * i, 16);
        v = e_add(v, g[i]);
        memcpy(nxt + 4 * i, v.c, 16);
      }
    }
    fri_layer[k] = layer;
    layer = nxt;
    shift_k = f_mul(shift_k, shift_k);
  }
  if (memcmp(layer, layer + 4, 16) != 0 && !getenv("ZKSP_ORACLE_FORCE")) return 3; /* some constraint does not hold */
  orc_ch_observe_many(&ch, layer, 4);
  put(&pb, layer, 4);
  free(layer);
  for (int l = 0; l < 32; ++l) free(G[l]);

  /* ---- proof of work, queries ---- */
  uint32_t witness = orc_ch_grind_padded(&ch, (int)cfg->pow_bits);
  put(&pb, &witness, 1);
  orc_ch_drop_outputs(&ch); /* (v16: the query indices start from a fresh squeeze) */
  const size_t hmax = (size_t)1 << lm;
  for (uint32_t q = 0; q < cfg->num_queries; ++q) {
    const size_t idx = orc_ch_sample_bits(&ch, lm + 1);
    const size_t cs = idx >> lm, m = idx & (hmax - 1);
    mmcs_open(cd, R_PREP, &t_prep, cs, m, &pb);
    mmcs_open(cd, R_MAIN, &t_main, cs, m, &pb);
    mmcs_open(cd, R_PERM, &t_perm, cs, m, &pb);
    mmcs_open(cd, R_QUOT, &t_quot, cs, m, &pb);
    for (int k = 0; k < lm; ++k) {
      const int loghk = lm - k;
      const size_t hk = (size_t)1 << loghk, half = hk >> 1, mk = m & (half - 1), leaf = cs * half + mk;
      put(&pb, fri_layer[k] + 4 * (cs * hk + mk), 4);
      put(&pb, fri_layer[k] + 4 * (cs * hk + mk + half), 4);
      for (int l = 0; l < loghk; ++l) put(&pb, fri_tree[k] + 8 * (orc_merkle_layer_offset(loghk, l) + ((leaf >> l) ^ 1)), 8);
    }
  }
  for (int k = 0; k < lm; ++k) { free(fri_tree[k]); free(fri_layer[k]); }
  free(fri_tree);
  free(fri_layer);
  mmcs_free(&t_prep); mmcs_free(&t_main); mmcs_free(&t_perm); mmcs_free(&t_quot);
  free_chips(cd);
  if (pb.n * 4 != need) return 4;
  return 0;
}
