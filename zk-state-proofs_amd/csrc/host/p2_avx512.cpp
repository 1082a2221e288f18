// Poseidon2 (BabyBear, width 16, x^7, 4 + 13 + 4 rounds) for a host with AVX-512: a whole state in ONE 512-bit register, and
// up to four independent states in lockstep (32 registers hold them with room to spare).  The GPU boxes' hosts are EPYC 9575F
// (Zen 5: a full-width 512-bit data path), where this is what the verifier hashes with; p2_avx2.cpp is the form for hosts
// without, poseidon2.hpp's scalar form the one for hosts with neither.  Same function, same constants (P2Consts.ext /
// internal / diag, Montgomery form), every lane kept in [0, p): bit-identical results
// (tests/test_verifier.py::test_host_poseidon2_vector_matches_scalar, through zksp_host_poseidon2_permute).
//
// Compiled as plain C++ with -mavx512f for this file alone; cpu_features.cpp (compiled without) asks the CPU before anything here runs.
#include <immintrin.h>
#include <stdint.h>

namespace zksp {
namespace p2avx512 {

namespace {
constexpr uint32_t kP = 0x78000001u;
constexpr uint32_t kMu = 0x88000001u;  // p^-1 mod 2^32

inline __m512i add_mod(__m512i a, __m512i b, __m512i p) {
  const __m512i t = _mm512_add_epi32(a, b);  // < 2p < 2^32
  return _mm512_min_epu32(t, _mm512_sub_epi32(t, p));
}
// a * b / 2^32 mod p, lanes in [0, p)
inline __m512i mul_mod(__m512i a, __m512i b, __m512i p, __m512i mu) {
  const __m512i a_odd = _mm512_srli_epi64(a, 32), b_odd = _mm512_srli_epi64(b, 32);
  const __m512i pe = _mm512_mul_epu32(a, b), po = _mm512_mul_epu32(a_odd, b_odd);
  const __m512i qe = _mm512_mul_epu32(pe, mu), qo = _mm512_mul_epu32(po, mu);  // low words: q = T * p^-1 mod 2^32
  const __m512i qpe = _mm512_mul_epu32(qe, p), qpo = _mm512_mul_epu32(qo, p);
  // T - q p is a multiple of 2^32 in (-p 2^32, p 2^32): its high word is the result or the result - p
  const __m512i de = _mm512_sub_epi64(pe, qpe), dof = _mm512_sub_epi64(po, qpo);
  const __m512i hi = _mm512_mask_blend_epi32((__mmask16)0xaaaa, _mm512_srli_epi64(de, 32), dof);
  return _mm512_min_epu32(hi, _mm512_add_epi32(hi, p));
}
inline uint32_t mul_mod1(uint32_t a, uint32_t b) {
  const uint64_t t = (uint64_t)a * b;
  const uint32_t q = (uint32_t)t * kMu;
  const int64_t d = (int64_t)t - (int64_t)((uint64_t)q * kP);
  const int32_t hi = (int32_t)(d >> 32);
  return hi < 0 ? (uint32_t)(hi + (int32_t)kP) : (uint32_t)hi;
}
// circ(2 M4, M4, M4, M4), M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]: a 128-bit lane holds one block of four
inline __m512i external_linear(__m512i v, __m512i p) {
  const __m512i t = add_mod(v, _mm512_shuffle_epi32(v, (_MM_PERM_ENUM)0xb1), p);    // x0+x1 x0+x1 x2+x3 x2+x3
  const __m512i sum = add_mod(t, _mm512_shuffle_epi32(t, (_MM_PERM_ENUM)0x4e), p);  // the block's sum, in every word
  const __m512i r = _mm512_shuffle_epi32(v, (_MM_PERM_ENUM)0x39);                   // x1 x2 x3 x0
  const __m512i y = add_mod(add_mod(sum, v, p), add_mod(r, r, p), p);               // y_i = sum + x_i + 2 x_(i+1)
  const __m512i s = add_mod(y, _mm512_shuffle_i32x4(y, y, 0xb1), p);                // blocks 0+1 0+1 2+3 2+3
  const __m512i all = add_mod(s, _mm512_shuffle_i32x4(s, s, 0x4e), p);              // the four blocks' sum in every block
  return add_mod(y, all, p);
}

template <int N>
inline void permute_n(uint32_t* const* st, const uint32_t (*ext)[16], const uint32_t* internal, const uint32_t* diag) {
  const __m512i p = _mm512_set1_epi32((int)kP), mu = _mm512_set1_epi32((int)kMu);
  const __m512i d = _mm512_loadu_si512(diag), lo32 = _mm512_set1_epi64(0xffffffffll);
  __m512i v[N];
  for (int t = 0; t < N; ++t) v[t] = external_linear(_mm512_loadu_si512(st[t]), p);
  for (int r = 0; r < 8; ++r) {
    if (r == 4) {
      for (int ir = 0; ir < 13; ++ir) {
        uint32_t x[N], x2[N], x3[N], x4[N];
        for (int t = 0; t < N; ++t) {  // element 0: + constant, x^7 (N chains side by side)
          x[t] = (uint32_t)_mm512_cvtsi512_si32(v[t]) + internal[ir];
          x[t] = x[t] >= kP ? x[t] - kP : x[t];
        }
        for (int t = 0; t < N; ++t) x2[t] = mul_mod1(x[t], x[t]);
        for (int t = 0; t < N; ++t) x3[t] = mul_mod1(x2[t], x[t]);
        for (int t = 0; t < N; ++t) x4[t] = mul_mod1(x2[t], x2[t]);
        for (int t = 0; t < N; ++t) v[t] = _mm512_mask_mov_epi32(v[t], (__mmask16)1, _mm512_set1_epi32((int)mul_mod1(x3[t], x4[t])));
        __m512i sum[N];
        for (int t = 0; t < N; ++t) {  // the sum of the sixteen words (64-bit lanes: below 2^35), reduced once
          const __m512i q = _mm512_add_epi64(_mm512_and_si512(v[t], lo32), _mm512_srli_epi64(v[t], 32));
          const uint64_t total = (uint64_t)_mm512_reduce_add_epi64(q);
          sum[t] = _mm512_set1_epi32((int)(uint32_t)(total % kP));
        }
        for (int t = 0; t < N; ++t) v[t] = add_mod(mul_mod(v[t], d, p, mu), sum[t], p);
      }
    }
    const __m512i c = _mm512_loadu_si512(ext[r]);
    __m512i x2[N], x3[N], x4[N];
    for (int t = 0; t < N; ++t) v[t] = add_mod(v[t], c, p);
    for (int t = 0; t < N; ++t) x2[t] = mul_mod(v[t], v[t], p, mu);
    for (int t = 0; t < N; ++t) x3[t] = mul_mod(x2[t], v[t], p, mu);
    for (int t = 0; t < N; ++t) x4[t] = mul_mod(x2[t], x2[t], p, mu);
    for (int t = 0; t < N; ++t) v[t] = external_linear(mul_mod(x3[t], x4[t], p, mu), p);
  }
  for (int t = 0; t < N; ++t) _mm512_storeu_si512(st[t], v[t]);
}
}  // namespace

void permute1(uint32_t* a, const uint32_t (*ext)[16], const uint32_t* internal, const uint32_t* diag) {
  uint32_t* st[1] = {a};
  permute_n<1>(st, ext, internal, diag);
}
void permute2(uint32_t* a, uint32_t* b, const uint32_t (*ext)[16], const uint32_t* internal, const uint32_t* diag) {
  uint32_t* st[2] = {a, b};
  permute_n<2>(st, ext, internal, diag);
}
void permute4(uint32_t* a, uint32_t* b, uint32_t* c, uint32_t* d, const uint32_t (*ext)[16], const uint32_t* internal,
              const uint32_t* diag) {
  uint32_t* st[4] = {a, b, c, d};
  permute_n<4>(st, ext, internal, diag);
}

}  // namespace p2avx512
}  // namespace zksp
