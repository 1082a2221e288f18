// Machine-proof verifying key, header and verifier (see mverifier.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "machine.hpp"
#include "machine_defs.hpp"
#include "context.hpp"

namespace zksp {

// What `client.setup(ELF)` (reference prover/src/bin/main.rs:70) yields for the machine proof: the
// commitment to the preprocessed Program and Image tables, the entry point and table heights, and
// a digest binding all of it (observed first in the transcript).
struct MachineVk {
  uint32_t prep_root[8];  // canonical
  uint32_t digest[8];     // canonical
  uint32_t entry;
  uint32_t pad_pc;        // the padding instruction (last Program row): where HALT goes and where every proof ends
  int log_prog, log_image, keccak_mode;
};

struct MachineHeader {
  int logh[mach::kNumChips];
  uint32_t exit_code, pv_len;
  uint32_t handover_pc[mach::kNumCpuInst - 1];  // the pc CPU instance i + 1 starts at
  uint32_t pv_digest[8], deferred_digest[8], vk_digest[8];
  uint32_t agg_n, agg_root[8], agg_digest[8];  // aggregation payload: leaf count (0: none), Merkle root, digest of the leaf list
  uint32_t pub_n, pub_digest[8];               // public bus tuples (a leaf-proof check's statement): count, digest of the list
  size_t pv_offset, body_offset;
};

// preprocessed traces, canonical, column-major: image [4][2^log_image], program [12][2^log_prog], table [3][2^16]
void machine_prep_traces(const MachineProgram& prog, std::vector<uint32_t>* image_prep, std::vector<uint32_t>* program_prep,
                         std::vector<uint32_t>* table_prep);
// host-side commitment of the preprocessed tables (setup; no GPU)
void machine_host_setup(const MachineProgram& prog, MachineVk* vk);
size_t machine_proof_body_words(const int* logh, uint32_t num_queries);
// Coefficient of every opened value in the reduced openings (format v16): delta^(desc >> 16) * alpha_f^(desc & 0xffff), in the
// order the opened values are laid out (per chip: prep, main, perm, quot at zeta, then main, perm at zeta w)
void machine_reduce_exponents(const int* logh, std::vector<uint32_t>* desc);
void machine_reduce_coefs(const int* logh, const Fp4& af, const Fp4& delta, Fp4* out);
bool parse_machine_header(const uint8_t* bytes, size_t len, MachineHeader* h, std::string* err);
// 0 = accepted; 7 = malformed; 8 = rejected.  agg_leaves / n_agg: the leaves ([n][8] canonical words) of the aggregation
// payload the proof must carry (n_agg = 0: it must carry none).
// pub_tuples / n_pub: the public bus tuples the proof must close its buses with (n_pub = 0: it must carry none).  log: if
// given, receives the leaf-check records of THIS proof's transcript and query phase (so that another proof can establish
// them) and the public tuples such a proof closes its buses with (log->leaf_index names this proof among the leaves checked
// beside one run).  stub: `bytes` is a proof cut off in front of its query phase (machine_proof_body_words(logh, 0) body words):
// everything up to the proof of work is checked - header, transcript, bus balance, the constraint identity at zeta - the
// queries are not (a proof about them does that), and a log receives the public tuples only.
int verify_machine_proof(const uint8_t* bytes, size_t len, const MachineVk& vk, uint32_t num_queries, uint32_t pow_bits,
                         std::string* err, const uint32_t* agg_leaves = nullptr, size_t n_agg = 0, const uint32_t* agg_keys = nullptr,
                         const uint32_t* pub_tuples = nullptr, size_t n_pub = 0, LeafCheckLog* log = nullptr, bool stub = false,
                         unsigned max_threads = 0 /* of this call: 0 = up to eight */, struct ZetaSelfTest* zeta_selftest = nullptr);
// verify_machine_proof with this also runs the constraint identity at zeta as the recorded program (zeta_program.hpp) on the
// proof's values and compares every chip's folded constraints, and the final combination, with the native evaluation
struct ZetaSelfTest {
  uint32_t n_ops = 0, n_cells = 0, n_inputs = 0, n_consts = 0, max_reads = 0, inputs_read = 0;
  int mismatch_chip = -1;  // the first chip whose program value differs (-1: none; kNumChips: the final combination)
};
// sponge digest of a list of public bus tuples (what stands for the list in the proof header and the transcript)
void machine_pub_digest(const uint32_t* pub_tuples, size_t n_pub, uint32_t digest[8]);
// The aggregation payload's public part and the heap of digests the Poseidon2 chip's rows are expanded from:
// heap[8 k ..] = node k (root 1, children 2k and 2k + 1, leaf i at n + i; node 0 unused), canonical words.
// Aggregation payload: n digests supplied at heap keys (keys == nullptr: n + j, the leaves of a full tree; otherwise e.g. a
// leaf and the siblings along its Merkle path).  Root (node 1), the digest of the list (keys and digests) that stands for
// it in the transcript, and the Poseidon2 chip's rows: every ancestor of a supplied key in ascending order, as node records
// (mach::kP2RecWords words each: kind, tag 0, key, mask 0, left child's digest, right child's digest).  False if the set is malformed: a repeated or out-of-range key, a
// non-canonical word, an ancestor that is itself supplied or lacks a child.
bool machine_nodes_public(const uint32_t* keys, const uint32_t* digests, size_t n, uint32_t root[8], uint32_t list_digest[8],
                          std::vector<uint32_t>* rows);
// number of those rows without hashing anything (SIZE_MAX if malformed)
size_t machine_agg_row_count(const uint32_t* keys, size_t n);

}  // namespace zksp
