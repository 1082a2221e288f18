/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or called
 * from the product path (zk-state-proofs_amd/).  Only tests/, the smoke check and
 * bench.py's cpu_baseline leg may use it.
 *
 * PARITY UNPINNED against SP1/Plonky3: the arithmetic this file restates lives in
 * un-vendored third-party crates (p3-baby-bear / p3-field 0.1.4-succinct,
 * reference Cargo.lock:5157, :5239) whose sources and test vectors are not
 * available offline.  It restates the published
 * definitions: BabyBear p = 2^31 - 2^27 + 1, multiplicative generator 31,
 * quartic binomial extension F_p[x]/(x^4 - 11).  Pinned instead by algebraic
 * self-checks and an independent big-integer Python restatement
 * (tests/golden/gen_golden.py).
 *
 * Representation: canonical u32 in [0, p).  Deliberately the dumbest correct
 * thing (64-bit product, %), unlike the Montgomery form used on the device.
 */
#ifndef ZKSP_ORACLE_FIELD_H
#define ZKSP_ORACLE_FIELD_H
#include <stdint.h>

#define FP 2013265921u
#define F_GEN 31u
#define F_TWO_ADICITY 27
#define EXT_W 11u

typedef uint32_t fe;
typedef struct { fe c[4]; } fe4;

static inline fe f_add(fe a, fe b) { uint32_t s = a + b; return s >= FP ? s - FP : s; }
static inline fe f_sub(fe a, fe b) { return a >= b ? a - b : a + FP - b; }
static inline fe f_neg(fe a) { return a ? FP - a : 0; }
static inline fe f_mul(fe a, fe b) { return (fe)(((uint64_t)a * b) % FP); }
static inline fe f_pow(fe a, uint64_t e) {
  fe r = 1;
  while (e) { if (e & 1) r = f_mul(r, a); a = f_mul(a, a); e >>= 1; }
  return r;
}
static inline fe f_inv(fe a) { return f_pow(a, FP - 2); }
/* primitive 2^k-th root of unity */
static inline fe f_root_of_unity(int logn) { return f_pow(F_GEN, (uint64_t)(FP - 1) >> logn); }

static inline fe4 e_zero(void) { fe4 r = {{0, 0, 0, 0}}; return r; }
static inline fe4 e_one(void) { fe4 r = {{1, 0, 0, 0}}; return r; }
static inline fe4 e_from(fe a) { fe4 r = {{a, 0, 0, 0}}; return r; }
static inline int e_eq(fe4 a, fe4 b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3]; }
static inline fe4 e_add(fe4 a, fe4 b) { fe4 r; for (int i = 0; i < 4; ++i) r.c[i] = f_add(a.c[i], b.c[i]); return r; }
static inline fe4 e_sub(fe4 a, fe4 b) { fe4 r; for (int i = 0; i < 4; ++i) r.c[i] = f_sub(a.c[i], b.c[i]); return r; }
static inline fe4 e_neg(fe4 a) { fe4 r; for (int i = 0; i < 4; ++i) r.c[i] = f_neg(a.c[i]); return r; }
static inline fe4 e_mul_base(fe4 a, fe b) { fe4 r; for (int i = 0; i < 4; ++i) r.c[i] = f_mul(a.c[i], b); return r; }
static inline fe4 e_mul(fe4 a, fe4 b) {
  /* schoolbook product reduced by x^4 = 11 */
  fe t[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) t[i + j] = f_add(t[i + j], f_mul(a.c[i], b.c[j]));
  fe4 r;
  for (int i = 0; i < 4; ++i) r.c[i] = t[i];
  for (int i = 4; i < 7; ++i) r.c[i - 4] = f_add(r.c[i - 4], f_mul(EXT_W, t[i]));
  return r;
}
static inline fe4 e_pow(fe4 a, uint64_t e) {
  fe4 r = e_one();
  while (e) { if (e & 1) r = e_mul(r, a); a = e_mul(a, a); e >>= 1; }
  return r;
}
/* Frobenius x -> x^p : coefficient i is scaled by (11^((p-1)/4))^i */
static inline fe4 e_frob(fe4 a) {
  fe z = f_pow(EXT_W, (FP - 1) / 4), zi = 1;
  fe4 r;
  for (int i = 0; i < 4; ++i) { r.c[i] = f_mul(a.c[i], zi); zi = f_mul(zi, z); }
  return r;
}
/* a^-1 = (a^p a^{p^2} a^{p^3}) / Norm(a) */
static inline fe4 e_inv(fe4 a) {
  fe4 f1 = e_frob(a), f2 = e_frob(f1), f3 = e_frob(f2);
  fe4 t = e_mul(e_mul(f1, f2), f3);
  fe4 n = e_mul(a, t); /* lies in the base field */
  return e_mul_base(t, f_inv(n.c[0]));
}
#endif
