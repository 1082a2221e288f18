/* zksp_component.h - the round-1 keccak-chip COMPONENT path (proof format v2), kept as a kernel benchmark
 * (bench.py "keccak_chip_component") and as a test vehicle (tests/test_gpu_prover.py, tests/test_verifier.py).
 *
 * NOT part of the drop-in surface (include/zksp.h): a component proof establishes only that a list of keccak-f outputs
 * belongs to a list of inputs - it is not a proof of the guest's execution - and only a client created with
 * proof_mode = ZKSP_PROOF_KECCAK_CHIP makes or accepts one; a default client answers it with "not a machine proof".
 * Nothing a reference-side binding needs is declared here.
 *
 * BUILD SWITCH (round 5): the default libzksp.so does not contain this path.  `ZKSP_COMPONENT=1` in the environment of
 * zk-state-proofs_amd/build.py (and of the tests / bench that want it) builds and loads libzksp_component.so - the same
 * sources with -DZKSP_COMPONENT, plus prover.cpp / verifier.cpp and the keccak-only kernels. */
#ifndef ZKSP_COMPONENT_H
#define ZKSP_COMPONENT_H
#include "zksp.h"
#ifdef __cplusplus
extern "C" {
#endif

size_t zksp_proof_body_words(const zksp_client* c, int log_h);

/* Uploads a batch of keccak-f permutation inputs and transcript headers into the
 * client's HBM workspace: states [n][max_perms][25] u64 (row-major), n_perms [n],
 * init_obs [n][44] canonical u32 (vk digest, log_h, n_perms, exit code halves,
 * pv-digest halves, deferred-digest halves).  After this call the inputs are
 * resident; zksp_hip_prove_resident() can be timed on its own. */
int zksp_hip_load_batch(zksp_client* c, int log_h, size_t n, size_t max_perms, const uint64_t* states,
                        const uint32_t* n_perms, const uint32_t* init_obs);
/* Enqueues one full proving pass over the resident batch on the client's stream
 * (trace generation -> proof bodies in HBM).  Asynchronous. */
int zksp_hip_prove_resident(zksp_client* c);
/* Copies proof bodies [n][body_words] (canonical u32) to the host; synchronises. */
int zksp_hip_fetch_bodies(zksp_client* c, uint32_t* out, size_t cap_words);
/* Copies only the 8-word main-trace commitment of every resident proof ([n][8],
 * canonical u32): the 32 bytes per proof the multi-GPU farm all-gathers. */
int zksp_hip_fetch_roots(zksp_client* c, uint32_t* out, size_t cap_words);
/* Wraps one fetched body (zksp_hip_fetch_bodies) into a complete proof object: header,
 * public values and the public I/O list (input state and keccak-f of it per permutation)
 * are rebuilt from the same inputs zksp_hip_load_batch was given.  What zksp_prove does
 * for its own batches, exposed so that callers of the resident path (bench.py, tests) can
 * run zksp_verify on what they timed. */
int zksp_proof_from_body(const uint32_t* body, size_t body_words, uint32_t log_h, const uint64_t* states, uint32_t n_perms,
                         uint32_t exit_code, const uint8_t* public_values, size_t pv_len, const uint32_t* pv_digest,
                         const uint32_t* deferred_digest, const uint32_t* vk_digest, zksp_proof** out);

/* ---- kernel-level parity entry points of the component path's keccak-only kernels ---- */
/* row a3: states [n_perms][25] u64 -> trace [2633][2^log_h] */
int zksp_hip_keccak_trace(zksp_client* c, const uint64_t* d_states, uint32_t n_perms, int log_h, uint32_t* d_trace);
/* row a6: lde [2633][2][H], running-sum lde_p [4][2][H], challenges = alpha, gamma, beta,
 * cumulative sum (4 canonical words each) -> quotient values [8][H] */
int zksp_hip_keccak_quotient(zksp_client* c, const uint32_t* d_lde, const uint32_t* d_lde_p, int log_h,
                             const uint32_t* challenges, uint32_t* d_quot);
/* row a6, lookup argument: trace [2633][H], gamma_beta (8 canonical words) -> running sum
 * phi [4][H] and the cumulative sum (4 words) */
int zksp_hip_bus_perm_trace(zksp_client* c, const uint32_t* d_trace, int log_h, const uint32_t* gamma_beta, uint32_t* d_phi,
                            uint32_t* d_cum_sum);

#ifdef __cplusplus
}
#endif
#endif /* ZKSP_COMPONENT_H */
