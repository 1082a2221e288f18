// Machine-proof verifying key, header and verifier (see mverifier.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "machine.hpp"
#include "machine_defs.hpp"
#include "verifier.hpp"

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
  uint32_t exit_code, pv_len, handover_pc;
  uint32_t pv_digest[8], deferred_digest[8], vk_digest[8];
  size_t pv_offset, body_offset;
};

// preprocessed traces, canonical, column-major: image [4][2^log_image], program [12][2^log_prog], table [3][2^16]
void machine_prep_traces(const MachineProgram& prog, std::vector<uint32_t>* image_prep, std::vector<uint32_t>* program_prep,
                         std::vector<uint32_t>* table_prep);
// host-side commitment of the preprocessed tables (setup; no GPU)
void machine_host_setup(const MachineProgram& prog, MachineVk* vk);
size_t machine_proof_body_words(const int* logh, uint32_t num_queries);
bool parse_machine_header(const uint8_t* bytes, size_t len, MachineHeader* h, std::string* err);
// 0 = accepted; 7 = malformed; 8 = rejected
int verify_machine_proof(const uint8_t* bytes, size_t len, const MachineVk& vk, uint32_t num_queries, uint32_t pow_bits,
                         std::string* err);

}  // namespace zksp
