/* ORACLE -- TEST INFRASTRUCTURE ONLY (see field.h for the full header note).
 *
 * CPU restatement of the STARK hot path named by BASELINE.json's north_star
 * (SURVEY.md section 8a rows a3-a8).  The reference reaches this code only
 * through `client.prove(&pk, stdin).run()` (reference prover/src/bin/main.rs:71-74);
 * the implementation lives in un-vendored crates (sp1-* 3.4.0, p3-* 0.1.4-succinct,
 * reference Cargo.lock:5147-5397, :7083-7521).  PARITY UNPINNED vs SP1 proof
 * bytes: this restates the published algorithms (radix-2 NTT / coset LDE,
 * Poseidon2 width-16 sponge + 2-to-1 compression Merkle tree, keccak-f AIR in the
 * p3-keccak-air column layout, two-adic FRI, duplex-sponge Fiat-Shamir) under
 * this repository's own documented parameters (DESIGN.md "Proof format").
 */
#ifndef ZKSP_ORACLE_H
#define ZKSP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#include "field.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Poseidon2 (width 16, x^7, 8 external + 13 internal rounds) ---- */
#define P2_WIDTH 16
#define P2_RATE 8
#define P2_DIGEST 8
#define P2_EXT_ROUNDS 8
#define P2_INT_ROUNDS 13
void orc_poseidon2_constants(uint32_t* ext_rc /*[8*16]*/, uint32_t* int_rc /*[13]*/);
void orc_p2_external_linear(uint32_t* s /*[16]*/);
void orc_p2_internal_linear(uint32_t* s /*[16]*/);
void orc_poseidon2_permute(uint32_t* st /*[16]*/);
/* sponge hash of n field elements (overwrite mode, the last block zero-filled) */
void orc_hash_elems(const uint32_t* in, size_t n, uint32_t* out /*[8]*/);
void orc_compress(const uint32_t* l, const uint32_t* r, uint32_t* out /*[8]*/);

/* ---- NTT / LDE (p3-dft restated; natural order in and out) ---- */
void orc_ntt(uint32_t* a, int logn, int inverse);
void orc_dft_naive(const uint32_t* in, uint32_t* out, int logn);
/* in: ncols columns of H=2^logh evaluations over in_shift*K_H (column-major).
 * out: [ncols][2][H] evaluations over the two cosets g*K_H and g*w_{2H}*K_H
 * ("coset-major" LDE, blowup 2).  coefs (optional): [ncols][H] monomial coefficients. */
void orc_coset_lde(const uint32_t* in, int logh, int ncols, uint32_t in_shift, uint32_t* out, uint32_t* coefs);

/* ---- Merkle tree over the rows of a column-major matrix [W][N] ---- */
/* tree: 8*(2N-1) words: layer 0 = N leaf digests, layer 1 = N/2 ... root last */
void orc_merkle_commit(const uint32_t* mat, int width, int logn, uint32_t* tree);
size_t orc_merkle_layer_offset(int logn, int layer); /* in digests */

/* ---- duplex challenger ---- */
typedef struct {
  uint32_t state[16];
  uint32_t inbuf[8];
  int n_in;
  uint32_t outbuf[8];
  int n_out;
} orc_challenger;
void orc_ch_init(orc_challenger* c);
void orc_ch_observe(orc_challenger* c, uint32_t x);
void orc_ch_observe_many(orc_challenger* c, const uint32_t* x, size_t n);
uint32_t orc_ch_sample(orc_challenger* c);
void orc_ch_sample_ext(orc_challenger* c, uint32_t* out4);
uint32_t orc_ch_sample_bits(orc_challenger* c, int bits);
uint32_t orc_ch_grind(orc_challenger* c, int bits);
/* machine proofs since format v16: every phase of the transcript ends on a block boundary (a pending block is
 * zero-filled), so that a duplex is always "absorb eight words" or "squeeze" - the two row kinds of the in-circuit
 * transcript (machine.h "transcript chip") */
void orc_ch_pad(orc_challenger* c);          /* zero-fills and absorbs a pending block (no-op on a boundary) */
void orc_ch_drop_outputs(orc_challenger* c); /* the next sample starts from a fresh squeeze */
uint32_t orc_ch_grind_padded(orc_challenger* c, int bits); /* observe(w), pad, sample: the smallest such w */

/* ---- keccak-f[1600] AIR (p3-keccak-air column layout, 2633 columns) ---- */
#define KA_FLAGS 0
#define KA_EXPORT 24
#define KA_PREIMAGE 25
#define KA_A 125
#define KA_C 225
#define KA_CP 545
#define KA_AP 865
#define KA_APP 2465
#define KA_APP00 2565
#define KA_APPP00 2629
#define KA_WIDTH 2633
#define KA_NUM_CONSTRAINTS 3182     /* base-field AIR constraints */
#define KA_NUM_BUS_CONSTRAINTS 3    /* extension-valued LogUp constraints, indices 3182..3184 */
#define KA_BUS_TUPLE 200            /* input limbs (100) || output limbs (100) of one permutation */
#define KA_PERM_WIDTH 4             /* the running sum phi: one extension column = 4 base columns */
void orc_keccak_f(uint64_t* st /*[25]*/);
/* trace: column-major [KA_WIDTH][H], H = 2^logh >= 24*n_perms */
void orc_keccak_trace(const uint64_t* states_in, int n_perms, int logh, uint32_t* trace);
/* evaluate all constraints at one row pair; out[k] = c_k (base field) */
void orc_keccak_constraints(const uint32_t* local, const uint32_t* next, uint32_t is_first, uint32_t is_last,
                            uint32_t is_trans, uint32_t* out /*[KA_NUM_CONSTRAINTS]*/);
/* quotient values over the 2H-point LDE domain; lde: [W][2][H]; out: [8][H]
 * (column 4c+j = coefficient j of the extension value on coset c) */
void orc_keccak_quotient(const uint32_t* lde, int logh, const uint32_t* alpha4, uint32_t* out);

/* ---- LogUp bus: the chip RECEIVES (input limbs || output limbs) of every real
 * permutation with multiplicity `export`; the verifier sums the same tuples from the
 * public I/O list.  (Row a6 "lookup-argument constraints"; SP1 wires its precompile
 * chips to the CPU with the same argument.) ---- */
/* limbs of the public I/O list: out[p*200 + j], j < 100 input, j >= 100 keccak-f(input) */
void orc_bus_io_limbs(const uint64_t* states_in, int n_perms, uint32_t* out);
/* rows of the io matrix for a trace height (column-major [8][R], zero padded) */
int orc_bus_io_log_rows(int logh);
/* phi columns [4][H] (exclusive running sum of export/f over the trace rows) and the
 * cumulative sum S (4 words) for challenges gamma, beta */
void orc_bus_perm_trace(const uint32_t* trace, int logh, const uint32_t* gamma4, const uint32_t* beta4,
                        uint32_t* phi, uint32_t* cum_sum4);
/* sum over the public list of 1/(gamma + sum_j beta^j t_j): what S must equal */
void orc_bus_expected_sum(const uint32_t* io_limbs, int n_perms, const uint32_t* gamma4, const uint32_t* beta4,
                          uint32_t* out4);
/* quotient with the bus constraints: lde [W][2][H], lde_p [4][2][H] */
void orc_keccak_quotient_bus(const uint32_t* lde, const uint32_t* lde_p, int logh, const uint32_t* alpha4,
                             const uint32_t* gamma4, const uint32_t* beta4, const uint32_t* cum_sum4, uint32_t* out);

/* ---- FRI ---- */
/* one fold: in [2][Hk] ext (coset-major, 4 words per element) -> out [2][Hk/2] */
void orc_fri_fold(const uint32_t* in, int loghk, uint32_t shift_k, const uint32_t* beta4, uint32_t* out);

/* ---- whole proof ---- */
typedef struct {
  uint32_t log_h;
  uint32_t n_perms;
  uint32_t exit_code;
  uint32_t pv_len;
  uint32_t pv_digest[8];
  uint32_t deferred_digest[8];
  uint32_t vk_digest[8];
} orc_header;
typedef struct {
  uint32_t num_queries;
  uint32_t pow_bits;
} orc_config;
#define ZKSP_MAGIC 0x50534B5Au
#define ZKSP_VERSION 2u
size_t orc_proof_size(int logh, const orc_config* cfg, uint32_t pv_len, uint32_t n_perms);
size_t orc_proof_header_words(uint32_t pv_len, uint32_t n_perms);
/* returns 0 on success */
int orc_prove(const uint64_t* states_in, const orc_header* hdr, const uint8_t* public_values, const orc_config* cfg,
              uint8_t* out, size_t cap, size_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
