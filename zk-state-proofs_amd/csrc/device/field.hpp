// BabyBear (p = 2^31 - 2^27 + 1) in Montgomery form (R = 2^32) and its quartic
// binomial extension F_p[x]/(x^4 - 11).  Replaces p3-baby-bear / p3-field
// 0.1.4-succinct (reference Cargo.lock:5157, :5239) on the device; shared with the
// host verifier so both sides compute with one definition.
//
// All device buffers hold Montgomery residues in [0, p).  Canonical values only
// exist at the boundary (proof bytes, public inputs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zksp {

#define ZKSP_HD __host__ __device__ __forceinline__

constexpr uint32_t kP = 0x78000001u;        // 2013265921
constexpr uint32_t kMontyMu = 0x88000001u;  // p^-1 mod 2^32
constexpr uint32_t kMontyNegMu = 0x77ffffffu;  // -p^-1 mod 2^32
constexpr uint32_t kR1 = 0x0ffffffeu;       // 2^32 mod p  (Montgomery form of 1)
constexpr uint32_t kR2 = 0x45dddde3u;       // 2^64 mod p
constexpr uint32_t kGen = 31;               // multiplicative generator (canonical)
constexpr uint32_t kExtW = 11;              // x^4 = 11
constexpr uint32_t cmonty(uint32_t c) { return (uint32_t)((((uint64_t)c) << 32) % kP); }
constexpr uint32_t kExtWMonty = cmonty(kExtW);
constexpr uint32_t kFrobZMonty = cmonty(0x67055c21u);  // 11^((p-1)/4)

struct Fp {
  uint32_t v;  // Montgomery residue

  ZKSP_HD static Fp raw(uint32_t m) { Fp r; r.v = m; return r; }
  ZKSP_HD static Fp zero() { return raw(0); }
  ZKSP_HD static Fp one() { return raw(kR1); }

  // t < p * 2^32  ->  t * 2^-32 mod p
  ZKSP_HD static uint32_t reduce(uint64_t t) {
    // m*p == -t (mod 2^32): the low word of t + m*p cancels, the high word is the
    // result in [0, 2p).  One v_mul_lo_u32 + one v_mad_u64_u32 (measured on gfx950:
    // both run at about half the v_add_u32 rate, v_mul_hi_u32 at about 3/8).
    uint32_t m = (uint32_t)t * kMontyNegMu;
    uint64_t u = t + (uint64_t)m * kP;
    uint32_t r = (uint32_t)(u >> 32);
    uint32_t r2 = r - kP;
    return r < r2 ? r : r2;  // unsigned min: r - p wraps high exactly when r < p
  }
  ZKSP_HD static Fp from_canonical(uint32_t c) { return raw(reduce((uint64_t)c * kR2)); }
  ZKSP_HD uint32_t to_canonical() const { return reduce((uint64_t)v); }

  ZKSP_HD Fp operator+(Fp o) const {
    uint32_t s = v + o.v, t = s - kP;
    return raw(s < t ? s : t);
  }
  ZKSP_HD Fp operator-(Fp o) const {
    uint32_t d = v - o.v, t = d + kP;
    return raw(d < t ? d : t);
  }
  ZKSP_HD Fp operator-() const { return raw(v ? kP - v : 0); }
  ZKSP_HD Fp operator*(Fp o) const { return raw(reduce((uint64_t)v * o.v)); }
  ZKSP_HD Fp& operator+=(Fp o) { return *this = *this + o; }
  ZKSP_HD Fp& operator-=(Fp o) { return *this = *this - o; }
  ZKSP_HD Fp& operator*=(Fp o) { return *this = *this * o; }
  ZKSP_HD bool operator==(Fp o) const { return v == o.v; }
  ZKSP_HD bool operator!=(Fp o) const { return v != o.v; }
  ZKSP_HD Fp dbl() const { return *this + *this; }
  ZKSP_HD Fp sqr() const { return *this * *this; }

  ZKSP_HD Fp pow(uint64_t e) const {
    Fp r = one(), b = *this;
    while (e) {
      if (e & 1) r = r * b;
      b = b.sqr();
      e >>= 1;
    }
    return r;
  }
  ZKSP_HD Fp inv() const { return pow(kP - 2); }
};

ZKSP_HD Fp fp_root_of_unity(int logn) { return Fp::from_canonical(kGen).pow((uint64_t)(kP - 1) >> logn); }

struct Fp4 {
  Fp c[4];

  ZKSP_HD static Fp4 zero() { Fp4 r; r.c[0] = r.c[1] = r.c[2] = r.c[3] = Fp::zero(); return r; }
  ZKSP_HD static Fp4 one() { Fp4 r = zero(); r.c[0] = Fp::one(); return r; }
  ZKSP_HD static Fp4 from_base(Fp a) { Fp4 r = zero(); r.c[0] = a; return r; }

  ZKSP_HD Fp4 operator+(const Fp4& o) const { Fp4 r; for (int i = 0; i < 4; ++i) r.c[i] = c[i] + o.c[i]; return r; }
  ZKSP_HD Fp4 operator-(const Fp4& o) const { Fp4 r; for (int i = 0; i < 4; ++i) r.c[i] = c[i] - o.c[i]; return r; }
  ZKSP_HD Fp4 operator-() const { Fp4 r; for (int i = 0; i < 4; ++i) r.c[i] = -c[i]; return r; }
  ZKSP_HD Fp4 operator*(Fp b) const { Fp4 r; for (int i = 0; i < 4; ++i) r.c[i] = c[i] * b; return r; }
  ZKSP_HD Fp4 operator*(const Fp4& o) const;  // below, over the signed lazy layer
  ZKSP_HD Fp4& operator+=(const Fp4& o) { return *this = *this + o; }
  ZKSP_HD Fp4& operator-=(const Fp4& o) { return *this = *this - o; }
  ZKSP_HD Fp4& operator*=(const Fp4& o) { return *this = *this * o; }
  ZKSP_HD bool operator==(const Fp4& o) const { return c[0] == o.c[0] && c[1] == o.c[1] && c[2] == o.c[2] && c[3] == o.c[3]; }
  ZKSP_HD bool operator!=(const Fp4& o) const { return !(*this == o); }
  ZKSP_HD Fp4 sqr() const { return *this * *this; }
  ZKSP_HD Fp4 dbl() const { return *this + *this; }

  ZKSP_HD Fp4 pow(uint64_t e) const {
    Fp4 r = one(), b = *this;
    while (e) {
      if (e & 1) r = r * b;
      b = b.sqr();
      e >>= 1;
    }
    return r;
  }
  // Frobenius: coefficient i scaled by (11^((p-1)/4))^i
  ZKSP_HD Fp4 frob() const {
    const Fp z = Fp::raw(kFrobZMonty);
    Fp4 r;
    Fp zi = Fp::one();
    for (int i = 0; i < 4; ++i) { r.c[i] = c[i] * zi; zi = zi * z; }
    return r;
  }
  // a^-1 = (a^p a^{p^2} a^{p^3}) / Norm(a)
  ZKSP_HD Fp4 inv() const {
    Fp4 f1 = frob(), f2 = f1.frob(), f3 = f2.frob();
    Fp4 t = f1 * f2 * f3;
    Fp n = ((*this) * t).c[0];
    return t * n.inv();
  }
};

// ---------------------------------------------------------------------------
// Signed lazy arithmetic.  A "signed word" is ANY int32 congruent to a Montgomery-form
// value; canonical residues qualify.  The kernels are bound by vector-ALU issue and with
// p = 0.94 * 2^31 a canonical modular addition costs three instructions, so hot loops keep
// sums in 64-bit accumulators and reduce rarely:
//   fps_redc(T)        |T| < 0.5667 * 2^32 p : T / 2^32 mod p, |result| <= |T| / 2^32 + p/2
//                      (centred Montgomery reduction: m = lo(T) * (-p^-1) taken as signed)
//   fps_fold(T)        any |T| < 2^63        : T / 2^32 mod p, |result| < 0.57p
//                      (2^32 == c mod p, so T == hi * c + lo first)
//   fps_reduce_small(y) |y| < 2^37           : y mod p, |result| < 0.55p, one multiplication
//   fps_canon(t)       |t| < p               : the canonical residue
// A product of two signed words with |a|, |b| < 2^31 satisfies the fps_redc bound and
// comes back below 1.034p: closed under multiplication with no conditional subtraction.
// ---------------------------------------------------------------------------
constexpr uint32_t kRModP = kR1;  // c = 2^32 mod p = 268435454
constexpr int32_t fps_centre_const(uint32_t v) { return v > kP / 2 ? (int32_t)(v - kP) : (int32_t)v; }
constexpr int32_t kR2Centred = fps_centre_const(kR2);  // K

ZKSP_HD int32_t fps_redc(int64_t t) {
  const int32_t m = (int32_t)((uint32_t)t * kMontyNegMu);
  return (int32_t)((t + (int64_t)m * (int64_t)kP) >> 32);
}
ZKSP_HD int32_t fps_mul(int32_t a, int32_t b) { return fps_redc((int64_t)a * (int64_t)b); }
ZKSP_HD int32_t fps_fold(int64_t t) {
  const int32_t hi = (int32_t)(t >> 32);
  return fps_redc((int64_t)hi * (int64_t)kRModP + (int64_t)(uint32_t)t);
}
// |y| < 2^37 (a 64-bit sum of a few dozen words) -> y mod p as a signed word, |t| < 0.55p, with
// ONE multiplication: q = round(y / p) is estimated from the top bits, s = y >> 26 (|s| < 2^11),
// q = (s * 2185 + 2^15) >> 16 (2185 / 2^16 equals 2^26 / p to 2e-4, so |q - y/p| < 0.55), and
// t = y - q p is taken in 32 bits.  Same residue class as y (no factor of R involved).
ZKSP_HD int32_t fps_reduce_small(int64_t y) {
  const int32_t s = (int32_t)(y >> 26);
#if defined(__HIP_DEVICE_COMPILE__)
  const int32_t q = (__mul24(s, 2185) + 32768) >> 16;
#else
  const int32_t q = (s * 2185 + 32768) >> 16;
#endif
  return (int32_t)((uint32_t)y - (uint32_t)q * kP);
}
ZKSP_HD uint32_t fps_canon(int32_t t) {
  const uint32_t u = (uint32_t)t, w = u + kP;
  return u < w ? u : w;
}
// canonical residue -> the congruent word in (-p/2, p/2]
ZKSP_HD int32_t fps_centre(uint32_t v) { return v > kP / 2 ? (int32_t)(v - kP) : (int32_t)v; }

// Extension product x^4 = 11: every output coefficient is ONE 64-bit sum of four products of centred words,
// folded once (16 multiply-adds, 3 Montgomery products for 11 * b_k, 4 folds) instead of 19 canonical
// Montgomery products and 12 canonical additions.  |a|, |b| <= (p - 1) / 2 and |11 b_k| < 2^30.2, so a
// sum of four products stays below 2^62.2.  The result is the same field element, canonical.
ZKSP_HD Fp4 Fp4::operator*(const Fp4& o) const {
  const int32_t a0 = fps_centre(c[0].v), a1 = fps_centre(c[1].v), a2 = fps_centre(c[2].v), a3 = fps_centre(c[3].v);
  const int32_t b0 = fps_centre(o.c[0].v), b1 = fps_centre(o.c[1].v), b2 = fps_centre(o.c[2].v), b3 = fps_centre(o.c[3].v);
  constexpr int32_t kw = fps_centre_const(kExtWMonty);
  const int32_t w1 = fps_mul(kw, b1), w2 = fps_mul(kw, b2), w3 = fps_mul(kw, b3);
  const int64_t t0 = (int64_t)a0 * b0 + (int64_t)a1 * w3 + (int64_t)a2 * w2 + (int64_t)a3 * w1;
  const int64_t t1 = (int64_t)a0 * b1 + (int64_t)a1 * b0 + (int64_t)a2 * w3 + (int64_t)a3 * w2;
  const int64_t t2 = (int64_t)a0 * b2 + (int64_t)a1 * b1 + (int64_t)a2 * b0 + (int64_t)a3 * w3;
  const int64_t t3 = (int64_t)a0 * b3 + (int64_t)a1 * b2 + (int64_t)a2 * b1 + (int64_t)a3 * b0;
  Fp4 r;
  r.c[0] = Fp::raw(fps_canon(fps_fold(t0)));
  r.c[1] = Fp::raw(fps_canon(fps_fold(t1)));
  r.c[2] = Fp::raw(fps_canon(fps_fold(t2)));
  r.c[3] = Fp::raw(fps_canon(fps_fold(t3)));
  return r;
}

}  // namespace zksp
