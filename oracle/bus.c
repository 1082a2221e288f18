/* ORACLE -- TEST INFRASTRUCTURE ONLY (see field.h).
 *
 * LogUp bus of the keccak chip ("plus lookup-argument
 * constraints").  SP1 connects its precompile chips to the rest of the machine with a
 * LogUp permutation argument over an extension-field running-sum column
 * (sp1-stark / sp1-core-machine 3.4.0, reference Cargo.lock:7485, :7130); those sources
 * are absent, so this restates the published LogUp construction with this repository's
 * own tuple layout and constraint order.  PARITY UNPINNED vs SP1.
 *
 *   tuple of a trace row   t_j, j < 200: the 100 preimage limbs, then the 100 limbs of
 *                          the round's output state (a''' for lane 0, a'' for lanes 1..24)
 *   fingerprint            f = gamma + sum_j beta^j t_j
 *   multiplicity           m = export (1 on the last round of a real permutation)
 *   running sum            phi_0 = 0, phi_{i+1} = phi_i + m_i / f_i   (4 base columns)
 *   cumulative sum         S = phi_{H-1} + m_{H-1} / f_{H-1}
 *   constraints (ext)      L0 = is_first * phi
 *                          L1 = is_trans * ((phi_next - phi) * f - m)
 *                          L2 = is_last  * ((S - phi) * f - m)
 * The verifier recomputes S from the public I/O list.  Self-checks: tests/test_oracle.py.
 */
#include <stdlib.h>
#include <string.h>

#include "zksp_oracle.h"

void orc_bus_io_limbs(const uint64_t* states_in, int n_perms, uint32_t* out) {
  for (int p = 0; p < n_perms; ++p) {
    uint64_t st[25];
    memcpy(st, states_in + 25 * (size_t)p, sizeof st);
    for (int j = 0; j < 25; ++j)
      for (int l = 0; l < 4; ++l) out[(size_t)p * KA_BUS_TUPLE + 4 * j + l] = (uint32_t)((st[j] >> (16 * l)) & 0xffff);
    orc_keccak_f(st);
    for (int j = 0; j < 25; ++j)
      for (int l = 0; l < 4; ++l)
        out[(size_t)p * KA_BUS_TUPLE + 100 + 4 * j + l] = (uint32_t)((st[j] >> (16 * l)) & 0xffff);
  }
}

int orc_bus_io_log_rows(int logh) {
  size_t max_perms = ((size_t)1 << logh) / 24;
  size_t rows = (max_perms * KA_BUS_TUPLE + 7) / 8;
  int l = 0;
  while (((size_t)1 << l) < rows) ++l;
  return l;
}

/* column of tuple element j in a trace row */
static inline int tuple_col(int j) {
  if (j < 100) return KA_PREIMAGE + j;
  int o = j - 100, lane = o / 4, l = o % 4;
  return lane == 0 ? KA_APPP00 + l : KA_APP + 4 * lane + l;
}

static fe4 fingerprint(const uint32_t* row, fe4 gamma, const fe4* bpow) {
  fe4 f = gamma;
  for (int j = 0; j < KA_BUS_TUPLE; ++j) f = e_add(f, e_mul_base(bpow[j], row[tuple_col(j)]));
  return f;
}

static fe4* beta_powers(const uint32_t* beta4) {
  fe4 beta;
  memcpy(beta.c, beta4, 16);
  fe4* bpow = (fe4*)malloc(KA_BUS_TUPLE * sizeof(fe4));
  bpow[0] = e_one();
  for (int j = 1; j < KA_BUS_TUPLE; ++j) bpow[j] = e_mul(bpow[j - 1], beta);
  return bpow;
}

void orc_bus_perm_trace(const uint32_t* trace, int logh, const uint32_t* gamma4, const uint32_t* beta4, uint32_t* phi,
                        uint32_t* cum_sum4) {
  size_t h = (size_t)1 << logh;
  fe4 gamma;
  memcpy(gamma.c, gamma4, 16);
  fe4* bpow = beta_powers(beta4);
  uint32_t* row = (uint32_t*)malloc(KA_WIDTH * sizeof(uint32_t));
  fe4 acc = e_zero();
  for (size_t r = 0; r < h; ++r) {
    for (int j = 0; j < 4; ++j) phi[(size_t)j * h + r] = acc.c[j];
    for (int c = 0; c < KA_WIDTH; ++c) row[c] = trace[(size_t)c * h + r];
    fe4 f = fingerprint(row, gamma, bpow);
    acc = e_add(acc, e_mul_base(e_inv(f), row[KA_EXPORT]));
  }
  memcpy(cum_sum4, acc.c, 16);
  free(row);
  free(bpow);
}

void orc_bus_expected_sum(const uint32_t* io_limbs, int n_perms, const uint32_t* gamma4, const uint32_t* beta4,
                          uint32_t* out4) {
  fe4 gamma;
  memcpy(gamma.c, gamma4, 16);
  fe4* bpow = beta_powers(beta4);
  fe4 acc = e_zero();
  for (int p = 0; p < n_perms; ++p) {
    fe4 f = gamma;
    for (int j = 0; j < KA_BUS_TUPLE; ++j) f = e_add(f, e_mul_base(bpow[j], io_limbs[(size_t)p * KA_BUS_TUPLE + j]));
    acc = e_add(acc, e_inv(f));
  }
  memcpy(out4, acc.c, 16);
  free(bpow);
}

void orc_keccak_quotient_bus(const uint32_t* lde, const uint32_t* lde_p, int logh, const uint32_t* alpha4,
                             const uint32_t* gamma4, const uint32_t* beta4, const uint32_t* cum_sum4, uint32_t* out) {
  size_t h = (size_t)1 << logh;
  const int total = KA_NUM_CONSTRAINTS + KA_NUM_BUS_CONSTRAINTS;
  fe4 alpha, gamma, cum;
  memcpy(alpha.c, alpha4, 16);
  memcpy(gamma.c, gamma4, 16);
  memcpy(cum.c, cum_sum4, 16);
  fe4* apow = (fe4*)malloc((size_t)total * sizeof(fe4));
  apow[0] = e_one();
  for (int k = 1; k < total; ++k) apow[k] = e_mul(apow[k - 1], alpha);
  fe4* bpow = beta_powers(beta4);
  fe wh = f_root_of_unity(logh), w2h = f_root_of_unity(logh + 1);
  fe wh_inv = f_inv(wh);
  for (int c = 0; c < 2; ++c) {
    fe shift = c ? f_mul(F_GEN, w2h) : F_GEN;
    fe zh = f_sub(f_pow(shift, h), 1);
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
      fe4 phi, phi_next;
      for (int j = 0; j < 4; ++j) {
        phi.c[j] = lde_p[((size_t)j * 2 + c) * h + m];
        phi_next.c[j] = lde_p[((size_t)j * 2 + c) * h + mn];
      }
      fe x = f_mul(shift, f_pow(wh, m));
      fe is_first = f_mul(zh, f_inv(f_sub(x, 1)));
      fe is_last = f_mul(zh, f_inv(f_sub(x, wh_inv)));
      fe is_trans = f_sub(x, wh_inv);
      orc_keccak_constraints(local, next, is_first, is_last, is_trans, cons);
      fe4 acc = e_zero();
      for (int k = 0; k < KA_NUM_CONSTRAINTS; ++k) acc = e_add(acc, e_mul_base(apow[k], cons[k]));
      /* bus constraints (extension valued) */
      fe4 f = fingerprint(local, gamma, bpow);
      fe4 mult = e_from(local[KA_EXPORT]);
      fe4 l0 = e_mul_base(phi, is_first);
      fe4 l1 = e_mul_base(e_sub(e_mul(e_sub(phi_next, phi), f), mult), is_trans);
      fe4 l2 = e_mul_base(e_sub(e_mul(e_sub(cum, phi), f), mult), is_last);
      acc = e_add(acc, e_mul(apow[KA_NUM_CONSTRAINTS], l0));
      acc = e_add(acc, e_mul(apow[KA_NUM_CONSTRAINTS + 1], l1));
      acc = e_add(acc, e_mul(apow[KA_NUM_CONSTRAINTS + 2], l2));
      acc = e_mul_base(acc, zh_inv);
      for (int j = 0; j < 4; ++j) out[((size_t)(4 * c + j)) * h + m] = acc.c[j];
      free(local);
      free(cons);
    }
  }
  free(apow);
  free(bpow);
}
