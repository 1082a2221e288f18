/* ORACLE -- TEST INFRASTRUCTURE ONLY (see field.h).
 *
 * keccak-f[1600] AIR: trace generation, constraint evaluation and quotient
 * values.  Restates the published structure of p3-keccak-air 0.1.4-succinct
 * (reference Cargo.lock:5283; wrapped by sp1-core-machine, Cargo.lock:7130),
 * reached by the reference only beneath prover/src/bin/main.rs:71-74:
 * 24 rows per permutation, 2633 columns (step flags 24, export 1, preimage 100,
 * a 100, c 320, c' 320, a' 1600, a'' 100, a''[0][0] bits 64, a'''[0][0] limbs 4),
 * 64-bit lanes as 4 x u16 limbs, constraint degree 3.
 * PARITY UNPINNED vs Plonky3 (sources absent): constraint ORDER and the
 * ascending-alpha-power folding are this repository's own.
 * Pinned by: traces built from real keccak-f states satisfy every constraint,
 * single-cell corruptions violate at least one (tests/test_oracle.py).
 */
#include <stdlib.h>
#include <string.h>

#include "zksp_oracle.h"

static const uint64_t RC64[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
    0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
    0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
/* rotation offset of lane (x, y), indexed [x][y] */
static const int ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
static inline uint64_t rol64(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }

/* Fill one trace row for `round` given the round's input state `a` (lane x+5y)
 * and the permutation's preimage; advances `a` to the round's output. */
static void fill_row(uint32_t* row, const uint64_t* preimage, uint64_t* a, int round, int export_flag) {
  memset(row, 0, KA_WIDTH * sizeof(uint32_t));
  row[KA_FLAGS + round] = 1;
  row[KA_EXPORT] = (uint32_t)export_flag;
  for (int j = 0; j < 25; ++j)
    for (int l = 0; l < 4; ++l) {
      row[KA_PREIMAGE + 4 * j + l] = (uint32_t)((preimage[j] >> (16 * l)) & 0xffff);
      row[KA_A + 4 * j + l] = (uint32_t)((a[j] >> (16 * l)) & 0xffff);
    }
  uint64_t c[5], cp[5], ap[25], b[25], app[25];
  for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
  for (int x = 0; x < 5; ++x) cp[x] = c[x] ^ c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
  for (int j = 0; j < 25; ++j) ap[j] = a[j] ^ c[j % 5] ^ cp[j % 5];
  for (int x = 0; x < 5; ++x)
    for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(ap[x + 5 * y], ROT[x][y]);
  for (int y = 0; y < 5; ++y)
    for (int x = 0; x < 5; ++x) app[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
  uint64_t appp00 = app[0] ^ RC64[round];
  for (int x = 0; x < 5; ++x)
    for (int z = 0; z < 64; ++z) {
      row[KA_C + 64 * x + z] = (uint32_t)((c[x] >> z) & 1);
      row[KA_CP + 64 * x + z] = (uint32_t)((cp[x] >> z) & 1);
    }
  for (int j = 0; j < 25; ++j) {
    for (int z = 0; z < 64; ++z) row[KA_AP + 64 * j + z] = (uint32_t)((ap[j] >> z) & 1);
    for (int l = 0; l < 4; ++l) row[KA_APP + 4 * j + l] = (uint32_t)((app[j] >> (16 * l)) & 0xffff);
  }
  for (int z = 0; z < 64; ++z) row[KA_APP00 + z] = (uint32_t)((app[0] >> z) & 1);
  for (int l = 0; l < 4; ++l) row[KA_APPP00 + l] = (uint32_t)((appp00 >> (16 * l)) & 0xffff);
  memcpy(a, app, sizeof app);
  a[0] = appp00;
}

void orc_keccak_trace(const uint64_t* states_in, int n_perms, int logh, uint32_t* trace) {
  size_t h = (size_t)1 << logh;
  size_t total_perms = (h + 23) / 24;
#pragma omp parallel for schedule(static)
  for (size_t p = 0; p < total_perms; ++p) {
    uint64_t pre[25], a[25];
    uint32_t* row = (uint32_t*)malloc(KA_WIDTH * sizeof(uint32_t));
    if ((int)p < n_perms) memcpy(pre, states_in + 25 * p, sizeof pre);
    else memset(pre, 0, sizeof pre);
    memcpy(a, pre, sizeof a);
    for (int r = 0; r < 24; ++r) {
      size_t ri = p * 24 + (size_t)r;
      if (ri >= h) break;
      fill_row(row, pre, a, r, ((int)p < n_perms && r == 23) ? 1 : 0);
      for (int c = 0; c < KA_WIDTH; ++c) trace[(size_t)c * h + ri] = row[c];
    }
    free(row);
  }
}

static inline fe xor3(fe a, fe b, fe c) {
  /* a+b+c - 2(ab+ac+bc) + 4abc */
  fe ab = f_mul(a, b), ac = f_mul(a, c), bc = f_mul(b, c);
  fe s2 = f_add(f_add(ab, ac), bc);
  fe abc = f_mul(ab, c);
  fe r = f_add(f_add(a, b), c);
  r = f_sub(r, f_add(s2, s2));
  fe abc2 = f_add(abc, abc);
  return f_add(r, f_add(abc2, abc2));
}

/* bit z of B[X,Y] = rotl(A'[(X+3Y)%5, X], R[(X+3Y)%5][X]) */
static inline fe bbit(const uint32_t* local, int X, int Y, int z) {
  int xa = (X + 3 * Y) % 5, ya = X;
  int rot = ROT[xa][ya];
  return local[KA_AP + 64 * (5 * ya + xa) + ((z + 64 - rot) % 64)];
}

void orc_keccak_constraints(const uint32_t* local, const uint32_t* next, uint32_t is_first, uint32_t is_last,
                            uint32_t is_trans, uint32_t* out) {
  (void)is_last;
  int k = 0;
  const uint32_t* fl = local + KA_FLAGS;
  fe not_final = f_sub(1, fl[23]);
  fe trans_nf = f_mul(is_trans, not_final);
  /* --- MISC --- */
  for (int i = 0; i < 24; ++i) out[k++] = f_mul(is_first, i == 0 ? f_sub(fl[0], 1) : fl[i]);
  for (int i = 0; i < 24; ++i) out[k++] = f_mul(is_trans, f_sub(next[KA_FLAGS + (i + 1) % 24], fl[i]));
  for (int j = 0; j < 100; ++j) out[k++] = f_mul(fl[0], f_sub(local[KA_PREIMAGE + j], local[KA_A + j]));
  for (int j = 0; j < 100; ++j) out[k++] = f_mul(trans_nf, f_sub(next[KA_PREIMAGE + j], local[KA_PREIMAGE + j]));
  out[k++] = f_mul(local[KA_EXPORT], f_sub(local[KA_EXPORT], 1));
  out[k++] = f_mul(not_final, local[KA_EXPORT]);
  /* --- C(x) --- */
  for (int x = 0; x < 5; ++x)
    for (int z = 0; z < 64; ++z) {
      fe c = local[KA_C + 64 * x + z];
      out[k++] = f_mul(c, f_sub(c, 1));
      fe x3 = xor3(c, local[KA_C + 64 * ((x + 4) % 5) + z], local[KA_C + 64 * ((x + 1) % 5) + (z + 63) % 64]);
      out[k++] = f_sub(local[KA_CP + 64 * x + z], x3);
    }
  /* --- A(y,x) --- */
  for (int j = 0; j < 25; ++j) {
    int x = j % 5;
    for (int z = 0; z < 64; ++z) {
      fe v = local[KA_AP + 64 * j + z];
      out[k++] = f_mul(v, f_sub(v, 1));
    }
    for (int l = 0; l < 4; ++l) {
      fe acc = 0;
      for (int z = 16 * l + 15; z >= 16 * l; --z) {
        fe bit = xor3(local[KA_AP + 64 * j + z], local[KA_C + 64 * x + z], local[KA_CP + 64 * x + z]);
        acc = f_add(f_add(acc, acc), bit);
      }
      out[k++] = f_sub(local[KA_A + 4 * j + l], acc);
    }
  }
  /* --- P(x) --- */
  for (int x = 0; x < 5; ++x)
    for (int z = 0; z < 64; ++z) {
      fe s = 0;
      for (int y = 0; y < 5; ++y) s = f_add(s, local[KA_AP + 64 * (5 * y + x) + z]);
      fe d = f_sub(s, local[KA_CP + 64 * x + z]);
      out[k++] = f_mul(f_mul(d, f_sub(d, 2)), f_sub(d, 4));
    }
  /* --- CHI(y,x) --- */
  for (int j = 0; j < 25; ++j) {
    int X = j % 5, Y = j / 5;
    for (int l = 0; l < 4; ++l) {
      fe acc = 0;
      for (int z = 16 * l + 15; z >= 16 * l; --z) {
        fe b0 = bbit(local, X, Y, z), b1 = bbit(local, (X + 1) % 5, Y, z), b2 = bbit(local, (X + 2) % 5, Y, z);
        fe andn = f_mul(f_sub(1, b1), b2);
        fe t = f_mul(b0, andn);
        fe bit = f_sub(f_add(b0, andn), f_add(t, t));
        acc = f_add(f_add(acc, acc), bit);
      }
      out[k++] = f_sub(local[KA_APP + 4 * j + l], acc);
    }
  }
  /* --- IOTA --- */
  for (int z = 0; z < 64; ++z) {
    fe v = local[KA_APP00 + z];
    out[k++] = f_mul(v, f_sub(v, 1));
  }
  for (int l = 0; l < 4; ++l) {
    fe acc = 0;
    for (int z = 16 * l + 15; z >= 16 * l; --z) acc = f_add(f_add(acc, acc), local[KA_APP00 + z]);
    out[k++] = f_sub(local[KA_APP + l], acc);
  }
  for (int l = 0; l < 4; ++l) {
    fe acc = 0;
    for (int z = 16 * l + 15; z >= 16 * l; --z) {
      fe rc = 0;
      for (int r = 0; r < 24; ++r)
        if ((RC64[r] >> z) & 1) rc = f_add(rc, fl[r]);
      fe v = local[KA_APP00 + z];
      fe t = f_mul(v, rc);
      fe bit = f_sub(f_add(v, rc), f_add(t, t));
      acc = f_add(f_add(acc, acc), bit);
    }
    out[k++] = f_sub(local[KA_APPP00 + l], acc);
  }
  for (int j = 0; j < 25; ++j)
    for (int l = 0; l < 4; ++l) {
      fe o = (j == 0) ? local[KA_APPP00 + l] : local[KA_APP + 4 * j + l];
      out[k++] = f_mul(trans_nf, f_sub(next[KA_A + 4 * j + l], o));
    }
}

void orc_keccak_quotient(const uint32_t* lde, int logh, const uint32_t* alpha4, uint32_t* out) {
  size_t h = (size_t)1 << logh;
  fe4 alpha;
  memcpy(alpha.c, alpha4, 16);
  fe4* apow = (fe4*)malloc(KA_NUM_CONSTRAINTS * sizeof(fe4));
  apow[0] = e_one();
  for (int k = 1; k < KA_NUM_CONSTRAINTS; ++k) apow[k] = e_mul(apow[k - 1], alpha);
  fe wh = f_root_of_unity(logh), w2h = f_root_of_unity(logh + 1);
  fe wh_inv = f_inv(wh);
  for (int c = 0; c < 2; ++c) {
    fe shift = c ? f_mul(F_GEN, w2h) : F_GEN;
    fe zh = f_sub(f_pow(shift, h), 1); /* x^H - 1 is constant on the coset */
    fe zh_inv = f_inv(zh);
#pragma omp parallel for schedule(static)
    for (size_t m = 0; m < h; ++m) {
      uint32_t* local = (uint32_t*)malloc(2 * KA_WIDTH * sizeof(uint32_t));
      uint32_t* next = local + KA_WIDTH;
      uint32_t* cons = (uint32_t*)malloc(KA_NUM_CONSTRAINTS * sizeof(uint32_t));
      size_t mn = (m + 1) & (h - 1);
      for (int col = 0; col < KA_WIDTH; ++col) {
        local[col] = lde[((size_t)col * 2 + c) * h + m];
        next[col] = lde[((size_t)col * 2 + c) * h + mn];
      }
      fe x = f_mul(shift, f_pow(wh, m));
      fe is_first = f_mul(zh, f_inv(f_sub(x, 1)));
      fe is_last = f_mul(zh, f_inv(f_sub(x, wh_inv)));
      fe is_trans = f_sub(x, wh_inv);
      orc_keccak_constraints(local, next, is_first, is_last, is_trans, cons);
      fe4 acc = e_zero();
      for (int k = 0; k < KA_NUM_CONSTRAINTS; ++k) acc = e_add(acc, e_mul_base(apow[k], cons[k]));
      acc = e_mul_base(acc, zh_inv);
      for (int j = 0; j < 4; ++j) out[((size_t)(4 * c + j)) * h + m] = acc.c[j];
      free(local);
      free(cons);
    }
  }
  free(apow);
}
