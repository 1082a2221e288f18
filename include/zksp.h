/* zksp.h -- C ABI of the MI355X-native STARK prover for the reference's
 * sp1-merkle-proof guest.  Plain pointers and sizes only; no C++ or torch types.
 *
 * This is the drop-in boundary of SURVEY.md section 8b.  Each entry point names the
 * call of the reference's SP1 flow it replaces (reference prover/src/bin/main.rs,
 * test_generate_ethereum_transaction_zk_proof_sp1, lines 59-87).  The reference
 * binds sp1-sdk 3.4.0 (prover/Cargo.toml:12) at exactly these calls; a maintainer
 * switching backends binds these symbols instead (INTEGRATION.md shows the stub).
 *
 * Conventions: every function returns 0 on success or a ZKSP_ERR_* code; nothing
 * aborts or throws across the boundary; the caller owns every returned handle and
 * frees it with the matching *_free; byte pointers returned by accessors borrow
 * from their handle.  A client is bound to one GPU and one HIP stream and is not
 * re-entrant; use one client per GPU.
 */
#ifndef ZKSP_H
#define ZKSP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKSP_OK 0
#define ZKSP_ERR_INVALID_ARG 1
#define ZKSP_ERR_NO_DEVICE 2      /* HIP backend requested but no usable GPU */
#define ZKSP_ERR_HIP 3            /* a HIP runtime call failed */
#define ZKSP_ERR_ELF 4            /* setup(): not a loadable RV32IM ELF */
#define ZKSP_ERR_EXECUTOR 5       /* executor fault (illegal instruction, bad syscall ...) */
#define ZKSP_ERR_GUEST_PANIC 6    /* guest exited with a non-zero code (reference: panic) */
#define ZKSP_ERR_PROOF_FORMAT 7   /* deserialize(): malformed proof bytes */
#define ZKSP_ERR_VERIFY 8         /* verify(): proof rejected */
#define ZKSP_ERR_UNSUPPORTED 9

typedef struct zksp_client zksp_client;
typedef struct zksp_pk zksp_pk;
typedef struct zksp_vk zksp_vk;
typedef struct zksp_stdin zksp_stdin;
typedef struct zksp_proof zksp_proof;

/* keccak handling inside the executor (SURVEY.md section 0 finding 4 / section 7):
 * the committed guest runs software keccak-f; the prover needs the permutation
 * events either way. */
#define ZKSP_KECCAK_OBSERVE 1 /* run the guest's own keccakf, record its inputs (as-committed cycle counts) */
#define ZKSP_KECCAK_REPLACE 2 /* intercept keccakf as a precompile call (intended configuration) */

typedef struct {
  int32_t device_ordinal; /* HIP device index; -1 = verifier/executor only (no GPU touched) */
  int32_t keccak_mode;    /* ZKSP_KECCAK_*; 0 = default (REPLACE) */
  uint32_t num_queries;   /* FRI queries; 0 = default 100 */
  uint32_t pow_bits;      /* proof-of-work bits; 0xffffffff = default 16 */
  uint32_t max_batch;     /* most proofs proven in lockstep per launch group; 0 = default 192 (a chunk is also capped by
                             four fifths of the free HBM; a call starts with a first wave of an eighth of its runs and is cut
                             into at least two chunks behind it) */
  int32_t proof_mode;     /* ZKSP_PROOF_*; 0 = default (MACHINE) */
} zksp_options;
/* What zksp_prove / zksp_prove_batch establish:
 * MACHINE      the guest's whole execution (CPU instances, memory, program, ALU, keccak, multiplier ... chips joined by
 *              LogUp buses; proof format v16): the statement of the reference's client.prove().
 * KECCAK_CHIP  only "these keccak-f outputs belong to these inputs" (round-1 format v2, a component
 *              benchmark; NOT a proof that the guest ran).  A build switch: only the library built with ZKSP_COMPONENT=1
 *              (libzksp_component.so, zksp_component.h) has it; zksp_client_new of the default library answers
 *              ZKSP_ERR_UNSUPPORTED. */
#define ZKSP_PROOF_MACHINE 1
#define ZKSP_PROOF_KECCAK_CHIP 2
/* Layout of a serialized machine proof (zksp_proof_serialize; little-endian u32 words; csrc/host/machine_defs.hpp is the
 * source of these numbers - kMachineVersion, kHeaderWords):
 *   word 0           magic
 *   word 1           format version (16)
 *   words 2 .. 28    log2 height of each of the ZKSP_MACHINE_CHIPS = 27 chips, in proof order
 *   word 29, 30      guest exit code, length of the public values in bytes
 *   words 31 .. 54   sha256(public values) as 8 words, the deferred-proofs digest (8), the verifying-key digest (8)
 *   words 55 .. 61   the pc at which each of the seven later CPU instances starts (hand-over pcs)
 *   words 62 .. 78   aggregation payload: number of supplied digests (0: none), their Merkle root (8), digest of the list (8)
 *   words 79 .. 87   public bus tuples (a leaf-proof check's statement): their number (0: none), digest of the list (8)
 *   then             the public values, zero-padded to a word; then the proof body (commitment roots, cumulative sums,
 *                    opened values, FRI roots, final constant, proof-of-work witness, the query openings):
 *                    zksp_machine_body_words() words.  A proof STUB (zksp_proof_stub) ends in front of the query openings.
 * Everything in the header is absorbed into the transcript before the first challenge is drawn. */
#define ZKSP_MACHINE_HEADER_WORDS 88

/* replaces ProverClient::new()  (main.rs:61).  Fails with ZKSP_ERR_NO_DEVICE when
 * device_ordinal >= 0 and no GPU is usable: there is no CPU proving fallback.
 * Environment override ZKSP_PROVER (the counterpart of SP1_PROVER, reference
 * .env.example:1-2): "hip"/"local" = prove on the GPU, "host" = executor/verifier
 * only (as device_ordinal -1); any other value is ZKSP_ERR_UNSUPPORTED. */
int zksp_client_new(const zksp_options* opts, zksp_client** out);
void zksp_client_free(zksp_client* c);
/* last error text of this client ("" if none); never NULL */
const char* zksp_last_error(const zksp_client* c);

/* replaces client.setup(MERKLE_ELF) -> (pk, vk)  (main.rs:70) */
int zksp_setup(zksp_client* c, const uint8_t* elf, size_t elf_len, zksp_pk** pk, zksp_vk** vk);
void zksp_pk_free(zksp_pk* pk);
void zksp_vk_free(zksp_vk* vk);
/* 32-byte verifying-key digest (8 little-endian u32 field elements) */
int zksp_vk_digest(const zksp_vk* vk, const uint8_t** ptr, size_t* len);

/* replaces SP1Stdin::new() / stdin.write(&Vec<u8>)  (main.rs:62, :69).
 * write() applies the bincode Vec<u8> framing (u64-LE length prefix). */
zksp_stdin* zksp_stdin_new(void);
int zksp_stdin_write(zksp_stdin* s, const uint8_t* buf, size_t len);
void zksp_stdin_free(zksp_stdin* s);
/* Aggregation payload (stage 1; the reference's circuits/sp1-merkle-proof-recursive/src/main.rs:3-5
 * is a todo!()): besides the guest's run, the proof made from this stdin establishes the Poseidon2 Merkle root (2-to-1
 * compressions) of `n` digests of 8 canonical field words each - e.g. the main-trace commitments of n leaf proofs, as the
 * proof farm all-gathers them.  n: a power of two >= 2 (0 clears the payload).  The leaves are not part of the proof: the
 * verifier names them (zksp_verify_aggregate) and the proof carries their root (zksp_proof_aggregation). */
int zksp_stdin_set_aggregation(zksp_stdin* s, const uint32_t* leaves /* [n][8] */, size_t n);

/* Leaf-proof check (stage 2b: the transcript and the query phase of a verifier-in-circuit; the
 * reference's circuits/sp1-merkle-proof-recursive/src/main.rs:3-5 is a todo!(), its in-circuit verifier would be
 * sp1-recursion-*, Cargo.lock:7315-7417).  Besides the guest's run, the proof made from this stdin establishes that the QUERY
 * PHASE of `leaf` verifies UNDER THE CHALLENGES ITS OWN FIAT-SHAMIR TRANSCRIPT YIELDS: the transcript over the blocks named in
 * the statement (a transcript chip, one duplex per row); the canonical bits of every query's index word, from which every
 * opening's position follows (a query chip); for every FRI query, every opened row of the four commitment rounds hashed
 * (Poseidon2 sponge) and climbing its mixed-height Merkle path - injections included - to the root the transcript absorbed;
 * every FRI layer's sibling pair hashed and climbing to that layer's root; the reduced opening of every height (Horner sums
 * of the opened rows accumulated beside their hashes); and the folding chain from the tallest reduced opening through every
 * layer (shorter ones joining at their heights) to the final constant.  PUBLIC is a list of some 75 bus tuples
 * (zksp_leaf_public): the transcript's blocks, the preprocessed root, and a dozen constants derived from the challenges and
 * the values opened at zeta.  NOT in-circuit yet (checked by whoever derives that list, from the leaf or its STUB): the leaf's
 * bus balance and its constraint identity at zeta.  The call verifies `leaf` on the host first and fails (ZKSP_ERR_VERIFY) if it
 * does not verify: an honest prover has nothing to prove about a bad leaf.  leaf NULL clears every check of the stdin. */
int zksp_stdin_set_verified_leaf(zksp_client* c, zksp_stdin* s, const zksp_proof* leaf, const zksp_vk* leaf_vk);
/* ... and one more: the proof made from this stdin checks the query phases of ALL the leaves added so far (the node of a
 * recursion tree of that arity: config 5's 1024 leaf proofs are 256 such proofs of four leaves each, bench.py
 * tree_of_1024_leaves).  The k-th leaf's tags, root ids and tuples carry the leaf index k, so the checks share the Poseidon2,
 * query and transcript chips without sharing a tag; the statement is the leaves' public tuples one leaf after the other, in
 * this order (zksp_leaves_public, zksp_verify_with_leaves). */
int zksp_stdin_add_verified_leaf(zksp_client* c, zksp_stdin* s, const zksp_proof* leaf, const zksp_vk* leaf_vk);

/* replaces client.prove(&pk, stdin).run()  (main.rs:71-74).  `stdin` is consumed
 * (emptied) as in the reference; pk is borrowed.  A guest panic (reference:
 * `.expect("Failed to verify Merkle Proof")`, crypto-ops/src/lib.rs:20-22) returns
 * ZKSP_ERR_GUEST_PANIC with the guest's stderr text in zksp_last_error(). */
int zksp_prove(zksp_client* c, const zksp_pk* pk, zksp_stdin* stdin_, zksp_proof** out);
/* n independent proofs proven in lockstep on this client's GPU (throughput path of
 * BASELINE configs 3-5).  out[i] is NULL and status[i] != 0 for failed inputs. */
int zksp_prove_batch(zksp_client* c, const zksp_pk* pk, zksp_stdin* const* stdins, size_t n, zksp_proof** out,
                     int32_t* status);

/* replaces proof.public_values.to_vec()  (main.rs:75) */
int zksp_proof_public_values(const zksp_proof* p, const uint8_t** ptr, size_t* len);
int zksp_proof_serialize(const zksp_proof* p, const uint8_t** ptr, size_t* len);
int zksp_proof_deserialize(const uint8_t* buf, size_t len, zksp_proof** out);
void zksp_proof_free(zksp_proof* p);

/* replaces client.verify(&proof, &vk)  (main.rs:80).  Host-only; needs no GPU. */
int zksp_verify(zksp_client* c, const zksp_proof* p, const zksp_vk* vk);

/* ---- executor report (row a2; acceptance values in SURVEY.md appendix A.4) ---- */
typedef struct {
  uint64_t cycles;
  uint64_t memory_ops;
  uint32_t exit_code;
  uint32_t n_keccak;
  uint32_t pv_len;
  uint32_t pv_digest[8];
  uint64_t syscalls[6]; /* HALT, WRITE, COMMIT, COMMIT_DEFERRED, HINT_LEN, HINT_READ */
  uint64_t opcode_hist[64];
} zksp_exec_report;
/* Runs the guest only.  keccak_mode: 0 software, 1 observe, 2 replace.  Does not
 * consume stdin.  public_values (optional) receives up to pv_cap bytes; stderr_buf
 * (optional) the guest's fd-2 text, NUL terminated. */
int zksp_execute(zksp_client* c, const zksp_pk* pk, const zksp_stdin* stdin_, int keccak_mode, zksp_exec_report* report,
                 uint8_t* public_values, size_t pv_cap, char* stderr_buf, size_t stderr_cap);
const char* zksp_opcode_name(int index);
/* The keccak-f[1600] permutation inputs the guest run produces (25 u64 each, in call
 * order): exactly what the keccak chip's trace is generated from.  *n receives the
 * count; states (optional) up to cap_perms entries. */
int zksp_execute_keccak(zksp_client* c, const zksp_pk* pk, const zksp_stdin* stdin_, uint64_t* states, size_t cap_perms,
                        size_t* n);

/* ---- machine proof: traced execution records ----
 * The multi-chip proof binds a proof to the guest's execution: CPU, memory, program, keccak and
 * multiplier chips joined by LogUp buses.  These entry points expose what the chips are
 * generated from, as flat arrays (layouts: csrc/host/machine.hpp), for tests and the oracle. */
typedef struct zksp_mtrace zksp_mtrace;
#define ZKSP_MT_CYCLES 0        /* 12 u32 per executed cycle */
#define ZKSP_MT_KECCAK 1        /* 408 bytes per precompile call: ts, ptr, 25 u64 in, 50 u32 previous times */
#define ZKSP_MT_MEMFINAL 2      /* 5 u32 per image address and per other touched address: addr, init, fin, fin_ts, is_init (0: image word, 1: a word a
                                   HINT_READ covers - its initial value is the input's -, 2: any other address - it starts as zero) */
#define ZKSP_MT_MULS 3          /* 3 u32 per multiplier-chip row: kind (0 mul, 1 mulhu, 2 mulh, 3 mulhsu), b, c */
#define ZKSP_MT_PROG_MULT 4     /* u32 per Program-table row */
#define ZKSP_MT_ALU_IDX 5       /* u32 per ALU-chip row: index of the cycle (sll srl sra slt, blt bge) */
#define ZKSP_MT_PROGRAM 6       /* 9 u32 per row: pc, op, wr, use2, rd, rs1, rs2, imm, tgt; the last row is the padding instruction */
#define ZKSP_MT_IMAGE 7         /* 2 u32 per row: addr, value */
#define ZKSP_MT_PUBLIC_VALUES 8 /* bytes */
#define ZKSP_MT_SUB_IDX 9       /* u32 per sub-word-chip row: index of the cycle (lb lh lbu lhu sb sh) */
#define ZKSP_MT_BW_IDX 10       /* u32 per bitwise-chip row: index of the cycle (xor or and) */
#define ZKSP_MT_ECALL_IDX 11    /* u32 per ecall-chip row: index of the ecall cycle */
#define ZKSP_MT_LEAF_P2_ROWS 12   /* leaf-proof check: 32 u32 per Poseidon2-chip row (flags, tag, key, mask, 16 input words, root id,
                                     Horner sum, alpha_f) */
#define ZKSP_MT_LEAF_QR_ROWS 13   /* leaf-proof check: 132 u32 per query-chip row (the row itself, 31 rows per query) */
#define ZKSP_MT_LEAF_TR_ROWS 16   /* leaf-proof check: 32 u32 per transcript-chip row (flags, leaf, step, uses, 16 input words) */
#define ZKSP_MT_DIV_IDX 15         /* u32 per divider-chip row: index of the cycle (div divu rem remu) */
#define ZKSP_MT_LEAF_PUB_TUPLES 14 /* leaf-proof check: 16 u32 per public bus tuple (ZKSP_PUB_TUPLE_WORDS) */
typedef struct {
  uint64_t cycles;
  uint64_t memory_ops;
  uint32_t exit_code;
  uint32_t entry;
  uint32_t log_prog, log_image;
  uint32_t keccak_mode;
  uint32_t pv_digest[8];
  uint32_t deferred_digest[8];
  uint64_t uninit_reads; /* bytes-of-a-load events that read memory which was neither image, nor hinted, nor written before:
                            such memory starts with prover-chosen contents in the proof (SP1's treatment); 0 = the run cannot
                            be steered by them */
} zksp_mtrace_info_t;
/* Runs the guest with full tracing under the client's keccak mode (does not consume stdin). */
int zksp_machine_trace(zksp_client* c, const zksp_pk* pk, const zksp_stdin* stdin_, zksp_mtrace** out);
void zksp_mtrace_free(zksp_mtrace* t);
/* Borrowed pointer into the handle. */
int zksp_mtrace_section(const zksp_mtrace* t, int which, const void** ptr, size_t* bytes);
int zksp_mtrace_info(const zksp_mtrace* t, zksp_mtrace_info_t* info);
/* Device-resident machine proving (bench.py, parity tests): upload the records of n traced runs (they are proven
 * with one shape: zksp_machine_cover_heights), enqueue one proving pass, fetch the proof bodies ([n][body_words]
 * canonical u32; body_words = zksp_machine_body_words of that shape). */
#define ZKSP_MACHINE_CHIPS 27   /* cpu, keccak, keccak-mem, mem-final, image, program, mul, table, cpu2, alu, alu2, subword, subword2, bitwise, bitwise2, poseidon2, ecall, cpu3 .. cpu8, query, divider, transcript, hint */
int zksp_mtrace_heights(const zksp_mtrace* t, int32_t* log_heights /* [ZKSP_MACHINE_CHIPS] */);
size_t zksp_machine_body_words(const zksp_client* c, const int32_t* log_heights /* [ZKSP_MACHINE_CHIPS] */);
/* The shape a batch of these runs is proven with: the chip heights that cover the largest cycle / event / address
 * counts among them (zksp_hip_machine_load uses exactly this).  A run can be proven with any shape it fits. */
int zksp_machine_cover_heights(const zksp_mtrace* const* traces, size_t n, int32_t* log_heights /* [ZKSP_MACHINE_CHIPS] */);
/* Column widths of chip `chip` (0 .. ZKSP_MACHINE_CHIPS - 1, proof order): preprocessed, main, permutation (LogUp helper and
 * running-sum columns as base-field columns).  Returns the chip's name, NULL for an unknown chip.  (What bench.py prices its
 * algorithmic bytes with; sp1-core-machine's chip registry in SP1, reference Cargo.lock:7130.) */
const char* zksp_machine_chip_widths(int chip, int32_t* widths3);
int zksp_hip_machine_load(zksp_client* c, const zksp_pk* pk, const zksp_mtrace* const* traces, size_t n);
int zksp_hip_machine_prove(zksp_client* c);
/* Gives the client's batch workspace (one device arena, tens of gigabytes at large batches) and its pinned staging buffers
 * back; the next load or prove_batch allocates again.  For a process that holds several clients on one GPU. */
int zksp_hip_release_workspace(zksp_client* c);
int zksp_hip_machine_fetch_bodies(zksp_client* c, uint32_t* out, size_t cap_words);
/* Only the 8-word main-trace commitment of every resident proof ([n][8]): what the proof farm all-gathers. */
int zksp_hip_machine_fetch_roots(zksp_client* c, uint32_t* out, size_t cap_words);
/* The aggregation payload of a machine proof: leaf count (0: none) and the proven Merkle root. */
int zksp_proof_aggregation(const zksp_proof* p, uint32_t* n_leaves, uint32_t* root8);
/* client.verify for a proof with an aggregation payload: additionally, the root in the proof is the Poseidon2 Merkle root of
 * exactly these leaves.  (zksp_verify refuses such a proof: it cannot vouch for a root whose leaves it was not given.) */
int zksp_verify_aggregate(zksp_client* c, const zksp_proof* p, const zksp_vk* vk, const uint32_t* leaves /* [n][8] */, size_t n);
/* The same payload with the digests supplied at heap keys of the caller's choice (root 1, children 2K and 2K + 1; keys
 * below 2^30, none an ancestor of another, every ancestor with both children supplied or derived): keys NULL means n + j, the
 * leaves of a full tree; a leaf at key 2^d + i with sibling j at key ((2^d + i) >> j) ^ 1 makes the proof establish a
 * MERKLE PATH - "this leaf, hashed up along these siblings at position i, gives the header's root" (the building block of
 * an in-circuit FRI / MMCS verifier; reference stub: circuits/sp1-merkle-proof-recursive/src/main.rs:3-5). */
int zksp_stdin_set_aggregation_keyed(zksp_stdin* s, const uint32_t* keys /* [n] or NULL */, const uint32_t* digests /* [n][8] */, size_t n);
int zksp_verify_aggregate_keyed(zksp_client* c, const zksp_proof* p, const zksp_vk* vk, const uint32_t* keys /* [n] or NULL */,
                                const uint32_t* digests /* [n][8] */, size_t n);
/* The statement of a leaf-proof check (stage 2b) as the list of PUBLIC BUS TUPLES the proof's LogUp
 * buses close with (ZKSP_PUB_TUPLE_WORDS canonical u32 each: bus, 1 = the verifier sends it / 0 = receives it, multiplicity,
 * number of elements, up to 12 elements).  A proof made from a stdin with verified leaves establishes, for every leaf, that its
 * Fiat-Shamir transcript over the blocks named in the statement yields challenges under which EVERY FRI QUERY of the leaf
 * verifies: the index words drawn from the transcript, their canonical bits, the four mixed-height Merkle openings against
 * the roots the transcript absorbed, every FRI layer opening, the reduced openings from the opened rows, the folding chain
 * down to the final constant.  Per leaf the statement is: one TBLK tuple per absorbed block of the leaf's transcript (header
 * words, commitment roots, cumulative sums, the root of the values opened at zeta, FRI roots, final constant, witness) and
 * one TSQ tuple per squeeze; the preprocessed root (the key's); LEAFK and one BCONST per height (constants the verifier derives
 * from the challenges and the values opened at zeta); the proof-of-work word.  What is NOT in-circuit yet and is checked by
 * whoever derives the statement (zksp_leaf_public*): the leaf's bus balance and its constraint identity at zeta - for which
 * a PROOF STUB suffices (zksp_proof_stub: the proof without its query phase, about 5 % of its bytes).
 * zksp_leaf_public derives the list from a leaf proof or stub (checking everything but the queries on the way; out may be
 * NULL to size the buffer); zksp_verify_public checks a proof against a list; zksp_verify_with_leaf(s) does both.
 * zksp_verify refuses a proof that carries public tuples: it cannot vouch for a statement it was not given.  The proof header
 * holds the count and the sponge digest of the list (zksp_proof_public_tuples), which the transcript absorbs before any
 * challenge.  A leaf may itself check leaves (a NODE of a recursion tree): zksp_stdin_add_verified_node /
 * zksp_leaf_public_at take the node's own statement, whose digest its header carries. */
#define ZKSP_PUB_TUPLE_WORDS 16
int zksp_leaf_public(zksp_client* c, const zksp_proof* leaf, const zksp_vk* leaf_vk, uint32_t* out, size_t cap_words, size_t* n_tuples);
int zksp_verify_public(zksp_client* c, const zksp_proof* p, const zksp_vk* vk, const uint32_t* tuples, size_t n_tuples);
int zksp_verify_with_leaf(zksp_client* c, const zksp_proof* p, const zksp_vk* vk, const zksp_proof* leaf, const zksp_vk* leaf_vk);
/* the same for a proof that checks n leaves (zksp_stdin_add_verified_leaf), given in the order they were added */
int zksp_leaves_public(zksp_client* c, const zksp_proof* const* leaves, const zksp_vk* const* leaf_vks, size_t n, uint32_t* out,
                       size_t cap_words, size_t* n_tuples);
int zksp_verify_with_leaves(zksp_client* c, const zksp_proof* p, const zksp_vk* vk, const zksp_proof* const* leaves,
                            const zksp_vk* const* leaf_vks, size_t n);
int zksp_proof_public_tuples(const zksp_proof* p, uint32_t* n_tuples, uint32_t* digest8);
/* The proof without its query phase: header, public values, commitment roots, cumulative sums, the values opened at zeta, FRI
 * roots, final constant, witness.  What zksp_leaf_public* needs of a leaf; serialises like a proof. */
int zksp_proof_stub(const zksp_proof* p, zksp_proof** out);
/* zksp_leaf_public for the leaf at place `leaf_index` beside one run, which itself closes its buses with own_tuples (a node) */
int zksp_leaf_public_at(zksp_client* c, const zksp_proof* leaf_or_stub, const zksp_vk* leaf_vk, uint32_t leaf_index,
                        const uint32_t* own_tuples, size_t n_own_tuples, uint32_t* out, size_t cap_words, size_t* n_tuples);
/* Stage 2c, first piece: the constraint identity at zeta exists as a FIXED straight-line program of
 * operations c = a * b + d over extension cells, recorded from the same AIR templates the verifier evaluates natively
 * (csrc/host/zeta_program.hpp) - what an arithmetic chip will execute.  This verifies `proof_or_stub` (with its own statement
 * `own`, if it is a node) and, on the way, runs the program on the proof's opened values and compares every chip's folded
 * constraints and the final combination with the native evaluation.  info: [0] operations, [1] cells, [2] input cells,
 * [3] constant cells, [4] the first chip that disagrees (0xffffffff: none), [5] the largest number of reads of one cell, [6]
 * the input cells the program reads at all.  The run's memory argument - every cell written once, read as often as the program
 * says - is checked on the values as well. */
int zksp_zeta_program_selftest(zksp_client* c, const zksp_proof* proof_or_stub, const zksp_vk* vk, const uint32_t* own, size_t n_own,
                               uint32_t info[8]);
/* The same checks as zksp_stdin_add_verified_leaves, made LATER: by the zksp_prove / zksp_prove_batch call that consumes the stdin,
 * on its tracing threads, while the GPU proves the runs that are ready - so that the host's part of a recursion-tree level
 * (15 ms of every core per node of four leaves) runs beside the proving instead of in front of it.  The leaves, their keys
 * and the `own` tuples (copied) are only recorded here: the leaf proofs and keys must stay alive until that call returns.  A
 * leaf that does not verify makes the run's status ZKSP_ERR_VERIFY.  All of a run's leaves are deferred, or none. */
int zksp_stdin_defer_verified_leaves(zksp_client* c, zksp_stdin* s, const zksp_proof* const* leaves, const zksp_vk* const* leaf_vks,
                                     const uint32_t* const* own, const size_t* n_own, size_t n);
/* The statement of the proof that will be made from this stdin - the public bus tuples of the leaf checks it carries so far, in
 * the order they were added (zksp_stdin_add_verified_leaf(s) / _node computed them when they verified the leaves): what
 * zksp_leaves_public would derive again from the leaves.  n_tuples = 0 for a stdin without leaf checks.  A copy survives the
 * proving that consumes the checks (and is how the statement of DEFERRED checks is read: after the call that made them). */
int zksp_stdin_public_tuples(const zksp_stdin* s, uint32_t* out, size_t cap_words, size_t* n_tuples);
/* Several leaves at once, verified and logged side by side on the host's threads (a node of arity n in one call); own_tuples /
 * n_own_tuples: NULL, or per leaf the statement its proof was made for (NULL / 0 for a plain leaf) */
int zksp_stdin_add_verified_leaves(zksp_client* c, zksp_stdin* s, const zksp_proof* const* leaves, const zksp_vk* const* leaf_vks,
                                   const uint32_t* const* own_tuples, const size_t* n_own_tuples, size_t n);
/* zksp_stdin_add_verified_leaf for a leaf that is itself a node: own_tuples is the statement ITS proof was made for */
int zksp_stdin_add_verified_node(zksp_client* c, zksp_stdin* s, const zksp_proof* leaf, const zksp_vk* leaf_vk,
                                 const uint32_t* own_tuples, size_t n_own_tuples);
/* Kernel-level parity (tests): after zksp_hip_machine_prove, one intermediate matrix of resident proof `proof_index`, as
 * canonical u32, column-major [width][2^log_height]: stage 0 = a chip's main trace (trace expansion kernels; table chip:
 * the counted multiplicities), 1 = its LogUp permutation trace (helper columns + running sum), 2 = its quotient values
 * over the two cosets ([8][H]).  zksp_hip_machine_fetch_challenges: gamma, beta, alpha, zeta (4 words each), then the chips'
 * cumulative sums (4 words each), as the device's transcript sampled / computed them. */
int zksp_hip_machine_fetch_stage(zksp_client* c, int chip, int stage, size_t proof_index, uint32_t* out, size_t cap_words);
int zksp_hip_machine_fetch_challenges(zksp_client* c, size_t proof_index, uint32_t* out /* [16 + 4 * ZKSP_MACHINE_CHIPS] */);
/* Complete proof object (format v16) from one fetched body and the trace it belongs to.  log_heights: the shape the batch was
 * proven with (zksp_machine_cover_heights of the loaded traces), NULL = the trace's own minimal heights. */
int zksp_machine_proof_from_body(const zksp_pk* pk, const zksp_mtrace* t, const int32_t* log_heights /* [ZKSP_MACHINE_CHIPS] or NULL */,
                                 const uint32_t* body, size_t body_words, zksp_proof** out);
/* The machine-proof part of the verifying key: Merkle root of the preprocessed Program / Image
 * tables and the digest that binds it to the entry point, table heights and keccak mode
 * (8 canonical u32 each). */
int zksp_vk_machine(const zksp_vk* vk, uint32_t* prep_root8, uint32_t* digest8);

/* ---- device-resident hot path (bench.py, parity tests): zksp_hip_machine_* above; parameters, timing, profile ---- */
/* Proof-system parameters this build uses (for sizing buffers; the two widths are the keccak chip's). */
typedef struct {
  uint32_t trace_width;       /* 2633 */
  uint32_t num_constraints;   /* 3182 */
  uint32_t num_queries;
  uint32_t pow_bits;
  uint32_t max_batch;
} zksp_params;
int zksp_get_params(const zksp_client* c, zksp_params* out);
/* The round-1 keccak-chip COMPONENT proofs (format v2: "these keccak-f outputs belong to these inputs", not a proof of
 * execution) are a kernel benchmark and a test vehicle, not part of the drop-in surface: their entry points
 * (zksp_hip_load_batch, zksp_hip_prove_resident, zksp_hip_fetch_bodies, zksp_hip_fetch_roots, zksp_proof_from_body,
 * zksp_proof_body_words, and the parity entry points of its keccak-only kernels) are declared in zksp_component.h and
 * compiled only into libzksp_component.so (ZKSP_COMPONENT=1).  The default library neither makes nor accepts such proofs. */
int zksp_hip_sync(zksp_client* c);
/* HIP-event timing on the client's own stream. */
int zksp_hip_timer_start(zksp_client* c);
int zksp_hip_timer_stop(zksp_client* c, float* ms); /* synchronises */
/* Per-kernel HIP-event profile of zksp_hip_prove_resident (enable, run, read). */
int zksp_hip_profile_enable(zksp_client* c, int on);
int zksp_hip_profile_read(zksp_client* c, const char* kernel, double* total_ms, uint64_t* launches);
int zksp_hip_profile_reset(zksp_client* c);

/* ---- kernel-level entry points (parity tests; device pointers from
 * zksp_dev_malloc or any HIP allocation; field elements are Montgomery
 * residues x*2^32 mod p) ---- */
int zksp_dev_malloc(zksp_client* c, size_t bytes, void** out);
int zksp_dev_free(zksp_client* c, void* p);
int zksp_dev_upload(zksp_client* c, void* dst, const void* src, size_t bytes);
int zksp_dev_download(zksp_client* c, void* dst, const void* src, size_t bytes);
int zksp_dev_memset(zksp_client* c, void* dst, int value, size_t bytes);

/* row a4: in [ncols][H] evaluations over in_shift*K_H (in_shift canonical) ->
 * coefs_br [ncols][H] (optional, bit-reversed order) and lde [ncols][2][H] */
int zksp_hip_lde(zksp_client* c, const uint32_t* d_in, int log_h, size_t ncols, uint32_t in_shift, uint32_t* d_coefs_br,
                 uint32_t* d_lde);
/* row a5: mat [width][2^log_n] column-major -> tree [(2^(log_n+1)-1)][8] */
int zksp_hip_merkle_commit(zksp_client* c, const uint32_t* d_mat, int width, int log_n, uint32_t* d_tree);
int zksp_hip_poseidon2_permute(zksp_client* c, uint32_t* d_states, size_t n);
/* the HOST verifier's Poseidon2 permutation over n states of 16 canonical words, in place (no GPU): impl 0 = the scalar form,
 * 1 = the 256-bit vector form, 2 = that form on two states in lockstep (AVX2), 3 = a state per 512-bit register, 4 = four
 * such states in lockstep (AVX-512: how the verifier hashes the openings of four queries side by side) -
 * ZKSP_ERR_UNSUPPORTED where the CPU lacks the extension; all are the same function (csrc/host/p2_avx2.cpp, p2_avx512.cpp).
 * The verifier picks the widest form the CPU has; ZKSP_HOST_P2=scalar|avx2 in the environment lowers it. */
int zksp_host_poseidon2_permute(uint32_t* states, size_t n, int impl);
/* row a7: one FRI fold of layer [2][Hk][4] with challenge beta (canonical) */
int zksp_hip_fri_fold(zksp_client* c, const uint32_t* d_in, int log_hk, uint32_t shift_k, const uint32_t* beta,
                      uint32_t* d_out);
/* instruction-rate probe used by DESIGN.md's ALU roofline (which: 0 add, 1 mul_lo,
 * 2 mul_hi, 3 mad_u64_u32, 4 montgomery mul, 5 f64 fma, 100 64-bit shift-add) -> giga lane-ops/s;
 * 6 + k: Poseidon2 permutations per second with k + 1 workgroups per CU */
int zksp_hip_microbench(zksp_client* c, int which, double* gops);

#ifdef __cplusplus
}
#endif
#endif
