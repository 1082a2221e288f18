// Poseidon2 (BabyBear, width 16, x^7, 4 + 13 + 4 rounds) for the HOST's vector unit: one permutation in two 256-bit
// registers of eight canonical Montgomery words each.  The host verifier hashes as much as a proof's query phase opens
// (88 800 permutations per acct-d8 proof, 8 000 more for the list of opened values): with the scalar permutation
// (poseidon2.hpp, the device's signed lazy form run on a CPU core) that is 190 ms of one core per proof, and the leaf-proof
// check of a recursion-tree node waits for four of them.  Same function, same constants
// (P2Consts.ext / internal / diag, Montgomery form), every lane kept in [0, p): results are bit-identical to the scalar
// permutation (tests/test_verifier.py::test_host_poseidon2_vector_matches_scalar, through zksp_host_permute).
//
// Compiled as plain C++ with -mavx2 for this file alone; cpu_features.cpp (compiled without) asks the CPU before anything here runs.
#include <immintrin.h>
#include <stdint.h>

namespace zksp {
namespace p2avx2 {

namespace {
constexpr uint32_t kP = 0x78000001u;
constexpr uint32_t kMu = 0x88000001u;  // p^-1 mod 2^32

inline __m256i add_mod(__m256i a, __m256i b, __m256i p) {
  const __m256i t = _mm256_add_epi32(a, b);  // < 2p < 2^32
  return _mm256_min_epu32(t, _mm256_sub_epi32(t, p));
}
// a * b / 2^32 mod p, lanes in [0, p)
inline __m256i mul_mod(__m256i a, __m256i b, __m256i p, __m256i mu) {
  const __m256i a_odd = _mm256_srli_epi64(a, 32), b_odd = _mm256_srli_epi64(b, 32);
  const __m256i pe = _mm256_mul_epu32(a, b), po = _mm256_mul_epu32(a_odd, b_odd);
  const __m256i qe = _mm256_mul_epu32(pe, mu), qo = _mm256_mul_epu32(po, mu);  // low words: q = T * p^-1 mod 2^32
  const __m256i qpe = _mm256_mul_epu32(qe, p), qpo = _mm256_mul_epu32(qo, p);
  // T - q p is a multiple of 2^32 in (-p 2^32, p 2^32): its high word is the result or the result - p
  const __m256i de = _mm256_sub_epi64(pe, qpe), dof = _mm256_sub_epi64(po, qpo);
  const __m256i hi = _mm256_blend_epi32(_mm256_srli_epi64(de, 32), dof, 0xaa);
  return _mm256_min_epu32(hi, _mm256_add_epi32(hi, p));
}
inline __m256i sbox(__m256i x, __m256i p, __m256i mu) {
  const __m256i x2 = mul_mod(x, x, p, mu), x3 = mul_mod(x2, x, p, mu), x4 = mul_mod(x2, x2, p, mu);
  return mul_mod(x3, x4, p, mu);
}
inline uint32_t mul_mod1(uint32_t a, uint32_t b) {
  const uint64_t t = (uint64_t)a * b;
  const uint32_t q = (uint32_t)t * kMu;
  const int64_t d = (int64_t)t - (int64_t)((uint64_t)q * kP);
  const int32_t hi = (int32_t)(d >> 32);
  return hi < 0 ? (uint32_t)(hi + (int32_t)kP) : (uint32_t)hi;
}
// circ(2 M4, M4, M4, M4), M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]: a 128-bit lane holds one block of four
inline void external_linear(__m256i& v0, __m256i& v1, __m256i p) {
  auto m4 = [&](__m256i v) {
    const __m256i t = add_mod(v, _mm256_shuffle_epi32(v, 0xb1), p);    // x0+x1 x0+x1 x2+x3 x2+x3
    const __m256i sum = add_mod(t, _mm256_shuffle_epi32(t, 0x4e), p);  // the block's sum, in every word
    const __m256i r = _mm256_shuffle_epi32(v, 0x39);                   // x1 x2 x3 x0
    return add_mod(add_mod(sum, v, p), add_mod(r, r, p), p);           // y_i = sum + x_i + 2 x_(i+1)
  };
  const __m256i y0 = m4(v0), y1 = m4(v1);
  const __m256i s = add_mod(y0, y1, p);                                  // blocks 0+2 | 1+3
  const __m256i all = add_mod(s, _mm256_permute2x128_si256(s, s, 1), p);  // the four blocks' sum in both halves
  v0 = add_mod(y0, all, p);
  v1 = add_mod(y1, all, p);
}
}  // namespace

__attribute__((target("avx2"))) void permute(uint32_t* s, const uint32_t (*ext)[16], const uint32_t* internal, const uint32_t* diag) {
  const __m256i p = _mm256_set1_epi32((int)kP), mu = _mm256_set1_epi32((int)kMu);
  __m256i v0 = _mm256_loadu_si256((const __m256i*)s), v1 = _mm256_loadu_si256((const __m256i*)(s + 8));
  const __m256i d0 = _mm256_loadu_si256((const __m256i*)diag), d1 = _mm256_loadu_si256((const __m256i*)(diag + 8));
  external_linear(v0, v1, p);
  for (int r = 0; r < 4; ++r) {
    v0 = add_mod(v0, _mm256_loadu_si256((const __m256i*)ext[r]), p);
    v1 = add_mod(v1, _mm256_loadu_si256((const __m256i*)(ext[r] + 8)), p);
    v0 = sbox(v0, p, mu);
    v1 = sbox(v1, p, mu);
    external_linear(v0, v1, p);
  }
  const __m256i lo32 = _mm256_set1_epi64x(0xffffffffll);
  for (int r = 0; r < 13; ++r) {
    // element 0: + constant, x^7
    uint32_t x = (uint32_t)_mm256_cvtsi256_si32(v0) + internal[r];
    x = x >= kP ? x - kP : x;
    const uint32_t x2 = mul_mod1(x, x), x3 = mul_mod1(x2, x), x4 = mul_mod1(x2, x2);
    v0 = _mm256_blend_epi32(v0, _mm256_castsi128_si256(_mm_cvtsi32_si128((int)mul_mod1(x3, x4))), 1);
    // the sum of the sixteen words (64-bit lanes: below 2^35), reduced once
    const __m256i e = _mm256_add_epi64(_mm256_and_si256(v0, lo32), _mm256_and_si256(v1, lo32));
    const __m256i o = _mm256_add_epi64(_mm256_srli_epi64(v0, 32), _mm256_srli_epi64(v1, 32));
    const __m256i q = _mm256_add_epi64(e, o);
    const __m128i h = _mm_add_epi64(_mm256_castsi256_si128(q), _mm256_extracti128_si256(q, 1));
    const uint64_t total = (uint64_t)_mm_cvtsi128_si64(h) + (uint64_t)_mm_extract_epi64(h, 1);
    const __m256i sum = _mm256_set1_epi32((int)(uint32_t)(total % kP));
    v0 = add_mod(mul_mod(v0, d0, p, mu), sum, p);
    v1 = add_mod(mul_mod(v1, d1, p, mu), sum, p);
  }
  for (int r = 4; r < 8; ++r) {
    v0 = add_mod(v0, _mm256_loadu_si256((const __m256i*)ext[r]), p);
    v1 = add_mod(v1, _mm256_loadu_si256((const __m256i*)(ext[r] + 8)), p);
    v0 = sbox(v0, p, mu);
    v1 = sbox(v1, p, mu);
    external_linear(v0, v1, p);
  }
  _mm256_storeu_si256((__m256i*)s, v0);
  _mm256_storeu_si256((__m256i*)(s + 8), v1);
}

// Two independent permutations in lockstep: one permutation is a chain of dependent operations (every S-box waits for the
// linear layer before it, every internal round for the S-box of element 0 and the sum), which leaves most of the vector unit
// idle; a second state's chain fills the gaps.  The verifier hashes the openings of two queries side by side with this.
__attribute__((target("avx2"))) void permute2(uint32_t* sa, uint32_t* sb, const uint32_t (*ext)[16], const uint32_t* internal,
                                              const uint32_t* diag) {
  const __m256i p = _mm256_set1_epi32((int)kP), mu = _mm256_set1_epi32((int)kMu);
  __m256i a0 = _mm256_loadu_si256((const __m256i*)sa), a1 = _mm256_loadu_si256((const __m256i*)(sa + 8));
  __m256i b0 = _mm256_loadu_si256((const __m256i*)sb), b1 = _mm256_loadu_si256((const __m256i*)(sb + 8));
  const __m256i d0 = _mm256_loadu_si256((const __m256i*)diag), d1 = _mm256_loadu_si256((const __m256i*)(diag + 8));
  external_linear(a0, a1, p);
  external_linear(b0, b1, p);
  for (int r = 0; r < 8; ++r) {
    if (r == 4) {
      const __m256i lo32 = _mm256_set1_epi64x(0xffffffffll);
      for (int ir = 0; ir < 13; ++ir) {
        uint32_t xa = (uint32_t)_mm256_cvtsi256_si32(a0) + internal[ir], xb = (uint32_t)_mm256_cvtsi256_si32(b0) + internal[ir];
        xa = xa >= kP ? xa - kP : xa;
        xb = xb >= kP ? xb - kP : xb;
        const uint32_t xa2 = mul_mod1(xa, xa), xb2 = mul_mod1(xb, xb);
        const uint32_t xa3 = mul_mod1(xa2, xa), xb3 = mul_mod1(xb2, xb);
        const uint32_t xa4 = mul_mod1(xa2, xa2), xb4 = mul_mod1(xb2, xb2);
        a0 = _mm256_blend_epi32(a0, _mm256_castsi128_si256(_mm_cvtsi32_si128((int)mul_mod1(xa3, xa4))), 1);
        b0 = _mm256_blend_epi32(b0, _mm256_castsi128_si256(_mm_cvtsi32_si128((int)mul_mod1(xb3, xb4))), 1);
        const __m256i qa = _mm256_add_epi64(_mm256_add_epi64(_mm256_and_si256(a0, lo32), _mm256_and_si256(a1, lo32)),
                                            _mm256_add_epi64(_mm256_srli_epi64(a0, 32), _mm256_srli_epi64(a1, 32)));
        const __m256i qb = _mm256_add_epi64(_mm256_add_epi64(_mm256_and_si256(b0, lo32), _mm256_and_si256(b1, lo32)),
                                            _mm256_add_epi64(_mm256_srli_epi64(b0, 32), _mm256_srli_epi64(b1, 32)));
        const __m128i ha = _mm_add_epi64(_mm256_castsi256_si128(qa), _mm256_extracti128_si256(qa, 1));
        const __m128i hb = _mm_add_epi64(_mm256_castsi256_si128(qb), _mm256_extracti128_si256(qb, 1));
        const uint64_t ta = (uint64_t)_mm_cvtsi128_si64(ha) + (uint64_t)_mm_extract_epi64(ha, 1);
        const uint64_t tb = (uint64_t)_mm_cvtsi128_si64(hb) + (uint64_t)_mm_extract_epi64(hb, 1);
        const __m256i suma = _mm256_set1_epi32((int)(uint32_t)(ta % kP)), sumb = _mm256_set1_epi32((int)(uint32_t)(tb % kP));
        a0 = add_mod(mul_mod(a0, d0, p, mu), suma, p);
        b0 = add_mod(mul_mod(b0, d0, p, mu), sumb, p);
        a1 = add_mod(mul_mod(a1, d1, p, mu), suma, p);
        b1 = add_mod(mul_mod(b1, d1, p, mu), sumb, p);
      }
    }
    const __m256i c0 = _mm256_loadu_si256((const __m256i*)ext[r]), c1 = _mm256_loadu_si256((const __m256i*)(ext[r] + 8));
    a0 = add_mod(a0, c0, p); b0 = add_mod(b0, c0, p);
    a1 = add_mod(a1, c1, p); b1 = add_mod(b1, c1, p);
    // x^7 of the four registers, stage by stage: four independent chains
    const __m256i a0_2 = mul_mod(a0, a0, p, mu), b0_2 = mul_mod(b0, b0, p, mu), a1_2 = mul_mod(a1, a1, p, mu), b1_2 = mul_mod(b1, b1, p, mu);
    const __m256i a0_3 = mul_mod(a0_2, a0, p, mu), b0_3 = mul_mod(b0_2, b0, p, mu), a1_3 = mul_mod(a1_2, a1, p, mu), b1_3 = mul_mod(b1_2, b1, p, mu);
    const __m256i a0_4 = mul_mod(a0_2, a0_2, p, mu), b0_4 = mul_mod(b0_2, b0_2, p, mu), a1_4 = mul_mod(a1_2, a1_2, p, mu), b1_4 = mul_mod(b1_2, b1_2, p, mu);
    a0 = mul_mod(a0_3, a0_4, p, mu); b0 = mul_mod(b0_3, b0_4, p, mu);
    a1 = mul_mod(a1_3, a1_4, p, mu); b1 = mul_mod(b1_3, b1_4, p, mu);
    external_linear(a0, a1, p);
    external_linear(b0, b1, p);
  }
  _mm256_storeu_si256((__m256i*)sa, a0);
  _mm256_storeu_si256((__m256i*)(sa + 8), a1);
  _mm256_storeu_si256((__m256i*)sb, b0);
  _mm256_storeu_si256((__m256i*)(sb + 8), b1);
}


}  // namespace p2avx2
}  // namespace zksp
