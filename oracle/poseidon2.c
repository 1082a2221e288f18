/* ORACLE -- TEST INFRASTRUCTURE ONLY (see field.h).
 *
 * Poseidon2 over BabyBear, width 16, S-box x^7, 4+4 external and 13 internal
 * rounds; sponge (rate 8, overwrite mode, the last block zero-filled; inputs have fixed lengths) and 2-to-1 truncated-
 * permutation compression; restates p3-poseidon2 / p3-symmetric 0.1.4-succinct
 * (reference Cargo.lock:5353, :5367), called by the reference only beneath
 * prover/src/bin/main.rs:71-74.
 *
 * PARITY UNPINNED: SP1's 141 round constants are not available offline
 * (SURVEY.md appendix B).  This repository's constants are the first 141
 * accepted 31-bit words (< p, little-endian u32 & 0x7fffffff, rejection sampled)
 * of SHAKE256("zksp/poseidon2/babybear/w16/v1"): 64 initial external, 13
 * internal, 64 terminal external.  They are NOT SP1's.
 *
 * Linear layers (published Poseidon2 construction, Plonky3 parameterisation):
 *   external: M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] on each 4-chunk, then
 *             every element adds the sum of the 4 elements in its residue class mod 4
 *             (= circ(2*M4, M4, M4, M4));
 *   internal: y_i = d_i * x_i + sum(x), d = [-2, 1, 2, 4, ..., 8192, 32768].
 */
#include <string.h>

#include "zksp_oracle.h"

static const uint64_t RC64[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
    0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
    0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
static const int ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
static inline uint64_t rol64(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }

void orc_keccak_f(uint64_t* a) {
  for (int rnd = 0; rnd < 24; ++rnd) {
    uint64_t c[5], d[5], b[25];
    for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
    for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
    for (int x = 0; x < 5; ++x)
      for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y], ROT[x][y]);
    for (int y = 0; y < 5; ++y)
      for (int x = 0; x < 5; ++x) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= RC64[rnd];
  }
}

static uint32_t g_ext_rc[P2_EXT_ROUNDS][16];
static uint32_t g_int_rc[P2_INT_ROUNDS];
static uint32_t g_int_diag[16];
static int g_init = 0;

static void init_constants(void) {
  if (g_init) return;
  /* SHAKE256: rate 136, domain suffix 0x1f */
  static const char tag[] = "zksp/poseidon2/babybear/w16/v1";
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
  orc_keccak_f(st);
  uint32_t vals[141];
  int got = 0, pos = 0;
  while (got < 141) {
    if (pos == 136) {
      orc_keccak_f(st);
      pos = 0;
    }
    uint32_t w;
    memcpy(&w, (const uint8_t*)st + pos, 4);
    pos += 4;
    w &= 0x7fffffffu;
    if (w < FP) vals[got++] = w;
  }
  int k = 0;
  for (int r = 0; r < 4; ++r)
    for (int i = 0; i < 16; ++i) g_ext_rc[r][i] = vals[k++];
  for (int r = 0; r < P2_INT_ROUNDS; ++r) g_int_rc[r] = vals[k++];
  for (int r = 4; r < 8; ++r)
    for (int i = 0; i < 16; ++i) g_ext_rc[r][i] = vals[k++];
  g_int_diag[0] = FP - 2;
  for (int i = 1; i < 15; ++i) g_int_diag[i] = 1u << (i - 1);
  g_int_diag[15] = 1u << 15;
  g_init = 1;
}

void orc_poseidon2_constants(uint32_t* ext_rc, uint32_t* int_rc) {
  init_constants();
  memcpy(ext_rc, g_ext_rc, sizeof g_ext_rc);
  memcpy(int_rc, g_int_rc, sizeof g_int_rc);
}

static inline fe sbox(fe x) {
  fe x2 = f_mul(x, x), x3 = f_mul(x2, x), x4 = f_mul(x2, x2);
  return f_mul(x3, x4);
}

static void external_linear(fe* s) {
  for (int c = 0; c < 4; ++c) {
    fe a = s[4 * c], b = s[4 * c + 1], cc = s[4 * c + 2], d = s[4 * c + 3];
    fe two = 2, three = 3;
    s[4 * c] = f_add(f_add(f_mul(two, a), f_mul(three, b)), f_add(cc, d));
    s[4 * c + 1] = f_add(f_add(a, f_mul(two, b)), f_add(f_mul(three, cc), d));
    s[4 * c + 2] = f_add(f_add(a, b), f_add(f_mul(two, cc), f_mul(three, d)));
    s[4 * c + 3] = f_add(f_add(f_mul(three, a), b), f_add(cc, f_mul(two, d)));
  }
  fe sums[4];
  for (int j = 0; j < 4; ++j) sums[j] = f_add(f_add(s[j], s[4 + j]), f_add(s[8 + j], s[12 + j]));
  for (int i = 0; i < 16; ++i) s[i] = f_add(s[i], sums[i & 3]);
}

static void internal_linear(fe* s) {
  fe sum = 0;
  for (int i = 0; i < 16; ++i) sum = f_add(sum, s[i]);
  for (int i = 0; i < 16; ++i) s[i] = f_add(f_mul(s[i], g_int_diag[i]), sum);
}

/* the two linear layers, for the Poseidon2 chip of the machine proof (machine.c) */
void orc_p2_external_linear(uint32_t* s) { external_linear(s); }
void orc_p2_internal_linear(uint32_t* s) { init_constants(); internal_linear(s); }

void orc_poseidon2_permute(uint32_t* s) {
  init_constants();
  external_linear(s);
  for (int r = 0; r < 4; ++r) {
    for (int i = 0; i < 16; ++i) s[i] = sbox(f_add(s[i], g_ext_rc[r][i]));
    external_linear(s);
  }
  for (int r = 0; r < P2_INT_ROUNDS; ++r) {
    s[0] = sbox(f_add(s[0], g_int_rc[r]));
    internal_linear(s);
  }
  for (int r = 4; r < 8; ++r) {
    for (int i = 0; i < 16; ++i) s[i] = sbox(f_add(s[i], g_ext_rc[r][i]));
    external_linear(s);
  }
}

void orc_hash_elems(const uint32_t* in, size_t n, uint32_t* out) {
  uint32_t st[16];
  memset(st, 0, sizeof st);
  for (size_t off = 0; off < n; off += P2_RATE) {
    size_t m = n - off < P2_RATE ? n - off : P2_RATE;
    for (size_t i = 0; i < P2_RATE; ++i) st[i] = i < m ? in[off + i] : 0; /* overwrite mode; the last block is zero-filled */
    orc_poseidon2_permute(st);
  }
  memcpy(out, st, P2_DIGEST * sizeof(uint32_t));
}

void orc_compress(const uint32_t* l, const uint32_t* r, uint32_t* out) {
  uint32_t st[16];
  memcpy(st, l, 32);
  memcpy(st + 8, r, 32);
  orc_poseidon2_permute(st);
  memcpy(out, st, 32);
}

/* ---- duplex challenger (p3-challenger DuplexChallenger restated) ---- */
void orc_ch_init(orc_challenger* c) { memset(c, 0, sizeof *c); }

static void ch_duplex(orc_challenger* c) {
  for (int i = 0; i < c->n_in; ++i) c->state[i] = c->inbuf[i];
  c->n_in = 0;
  orc_poseidon2_permute(c->state);
  memcpy(c->outbuf, c->state, 32);
  c->n_out = 8;
}

void orc_ch_observe(orc_challenger* c, uint32_t x) {
  c->n_out = 0;
  c->inbuf[c->n_in++] = x;
  if (c->n_in == P2_RATE) ch_duplex(c);
}

void orc_ch_observe_many(orc_challenger* c, const uint32_t* x, size_t n) {
  for (size_t i = 0; i < n; ++i) orc_ch_observe(c, x[i]);
}

uint32_t orc_ch_sample(orc_challenger* c) {
  if (c->n_in != 0 || c->n_out == 0) ch_duplex(c);
  return c->outbuf[--c->n_out];
}

void orc_ch_sample_ext(orc_challenger* c, uint32_t* out4) {
  for (int i = 0; i < 4; ++i) out4[i] = orc_ch_sample(c);
}

uint32_t orc_ch_sample_bits(orc_challenger* c, int bits) { return orc_ch_sample(c) & ((1u << bits) - 1); }

uint32_t orc_ch_grind(orc_challenger* c, int bits) {
  for (uint32_t w = 0; w < FP; ++w) {
    orc_challenger t = *c;
    orc_ch_observe(&t, w);
    if (orc_ch_sample_bits(&t, bits) == 0) {
      *c = t;
      return w;
    }
  }
  return 0xffffffffu;
}

void orc_ch_pad(orc_challenger* c) {
  while (c->n_in != 0) orc_ch_observe(c, 0);
}
void orc_ch_drop_outputs(orc_challenger* c) { c->n_out = 0; }
uint32_t orc_ch_grind_padded(orc_challenger* c, int bits) {
  for (uint32_t w = 0; w < FP; ++w) {
    orc_challenger t = *c;
    orc_ch_observe(&t, w);
    orc_ch_pad(&t);
    if (orc_ch_sample_bits(&t, bits) == 0) {
      *c = t;
      return w;
    }
  }
  return 0xffffffffu;
}
