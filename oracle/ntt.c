/* ORACLE -- TEST INFRASTRUCTURE ONLY (see field.h).
 *
 * Radix-2 NTT and coset low-degree extension over BabyBear; restates p3-dft
 * 0.1.4-succinct (reference Cargo.lock:5226) as reached through sp1-stark
 * (Cargo.lock:7485) beneath reference prover/src/bin/main.rs:71-74.
 * PARITY UNPINNED vs Plonky3 (sources absent); pinned by the O(n^2) DFT below,
 * the inverse round trip and Horner evaluation on the coset (tests/test_oracle.py).
 *
 * Layout: column-major matrices (each column contiguous).  The LDE of a height-H
 * column with blowup 2 is stored "coset-major": out[col][c][m] = p(g * w_{2H}^c * w_H^m),
 * c in {0,1}, g = 31 the multiplicative generator.
 */
#include <stdlib.h>
#include <string.h>

#include "zksp_oracle.h"

static uint32_t bitrev(uint32_t v, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}

void orc_ntt(uint32_t* a, int logn, int inverse) {
  size_t n = (size_t)1 << logn;
  for (size_t i = 0; i < n; ++i) {
    size_t j = bitrev((uint32_t)i, logn);
    if (i < j) { uint32_t t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  for (int s = 1; s <= logn; ++s) {
    size_t m = (size_t)1 << s, half = m >> 1;
    fe wm = f_root_of_unity(s);
    if (inverse) wm = f_inv(wm);
    for (size_t k = 0; k < n; k += m) {
      fe w = 1;
      for (size_t j = 0; j < half; ++j) {
        fe t = f_mul(w, a[k + j + half]), u = a[k + j];
        a[k + j] = f_add(u, t);
        a[k + j + half] = f_sub(u, t);
        w = f_mul(w, wm);
      }
    }
  }
  if (inverse) {
    fe ninv = f_inv((fe)(n % FP));
    for (size_t i = 0; i < n; ++i) a[i] = f_mul(a[i], ninv);
  }
}

void orc_dft_naive(const uint32_t* in, uint32_t* out, int logn) {
  size_t n = (size_t)1 << logn;
  fe w = f_root_of_unity(logn);
  for (size_t i = 0; i < n; ++i) {
    fe wi = f_pow(w, i), x = 1, acc = 0;
    for (size_t k = 0; k < n; ++k) {
      acc = f_add(acc, f_mul(in[k], x));
      x = f_mul(x, wi);
    }
    out[i] = acc;
  }
}

void orc_coset_lde(const uint32_t* in, int logh, int ncols, uint32_t in_shift, uint32_t* out, uint32_t* coefs) {
  size_t h = (size_t)1 << logh;
  fe w2h = f_root_of_unity(logh + 1);
  fe shifts[2] = {F_GEN, f_mul(F_GEN, w2h)};
  fe in_shift_inv = f_inv(in_shift);
#pragma omp parallel for schedule(static)
  for (int col = 0; col < ncols; ++col) {
    uint32_t* c = (uint32_t*)malloc(h * sizeof(uint32_t));
    uint32_t* t = (uint32_t*)malloc(h * sizeof(uint32_t));
    memcpy(c, in + (size_t)col * h, h * sizeof(uint32_t));
    orc_ntt(c, logh, 1);
    /* evaluations were over in_shift*K_H: c_k currently holds coef_k * in_shift^k */
    fe s = 1;
    for (size_t k = 0; k < h; ++k) { c[k] = f_mul(c[k], s); s = f_mul(s, in_shift_inv); }
    if (coefs) memcpy(coefs + (size_t)col * h, c, h * sizeof(uint32_t));
    for (int cs = 0; cs < 2; ++cs) {
      fe p = 1;
      for (size_t k = 0; k < h; ++k) { t[k] = f_mul(c[k], p); p = f_mul(p, shifts[cs]); }
      orc_ntt(t, logh, 0);
      memcpy(out + ((size_t)col * 2 + cs) * h, t, h * sizeof(uint32_t));
    }
    free(c);
    free(t);
  }
}

/* ---- Merkle tree (p3-merkle-tree restated: one matrix, rows are leaves) ---- */
size_t orc_merkle_layer_offset(int logn, int layer) {
  size_t off = 0;
  for (int l = 0; l < layer; ++l) off += (size_t)1 << (logn - l);
  return off;
}

void orc_merkle_commit(const uint32_t* mat, int width, int logn, uint32_t* tree) {
  size_t n = (size_t)1 << logn;
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < n; ++r) {
    uint32_t* row = (uint32_t*)malloc((size_t)width * sizeof(uint32_t));
    for (int c = 0; c < width; ++c) row[c] = mat[(size_t)c * n + r];
    orc_hash_elems(row, (size_t)width, tree + 8 * r);
    free(row);
  }
  for (int l = 1; l <= logn; ++l) {
    size_t cnt = (size_t)1 << (logn - l);
    const uint32_t* prev = tree + 8 * orc_merkle_layer_offset(logn, l - 1);
    uint32_t* cur = tree + 8 * orc_merkle_layer_offset(logn, l);
#pragma omp parallel for schedule(static) if (cnt > 256)
    for (size_t i = 0; i < cnt; ++i) orc_compress(prev + 16 * i, prev + 16 * i + 8, cur + 8 * i);
  }
}
