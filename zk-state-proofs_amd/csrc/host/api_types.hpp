// Concrete layouts of the opaque C-ABI handles (include/zksp.h), shared by the
// translation units that implement the ABI.
#pragma once
#include <string>
#include <vector>

#include "context.hpp"
#include "machine.hpp"
#include "mverifier.hpp"
#ifdef ZKSP_COMPONENT
#include "verifier.hpp"
#endif

struct zksp_proof;
struct zksp_client { zksp::Context ctx; };
struct zksp_pk { zksp::ElfImage elf; uint32_t vk_digest[8]; zksp::MachineProgram mprog; zksp::MachineVk mvk; };
struct zksp_mtrace {
  zksp::MachineTrace t;
  int leaf_rc = 0;          // a deferred leaf check of this run failed (prove_batch): the code and what to say
  std::string leaf_err;
  const zksp::MachineProgram* prog;
  uint32_t handover_pc[zksp::mach::kNumCpuInst - 1] = {};  // pc of the first cycle of every later CPU instance (kept when the cycle records are released)
};
struct zksp_vk { uint32_t digest[8]; zksp::MachineVk machine; };
struct zksp_stdin {
  std::vector<std::vector<uint8_t>> entries;
  std::vector<uint32_t> agg_leaves;  // aggregation payload to prove beside the run (zksp_stdin_set_aggregation), 8 words per digest
  std::vector<uint32_t> agg_keys;    // heap keys of those digests (empty: the leaves of a full tree)
  std::shared_ptr<const zksp::LeafCheckLog> leaf_check;  // leaf-proof check to prove beside the run (zksp_stdin_set_verified_leaf)
  // leaf checks the next zksp_prove(_batch) call makes itself, on its tracing threads, while the GPU proves what is ready
  // (zksp_stdin_defer_verified_leaves): the leaves must outlive that call
  struct Deferred { const zksp_proof* leaf; const zksp_vk* vk; std::vector<uint32_t> own; };
  std::vector<Deferred> deferred;
  std::vector<uint32_t> statement;  // the public tuples of the leaf checks last attached (kept when proving consumes them)
};
// api_machine.cpp: runs a stdin's deferred leaf checks on at most `budget` threads (the leaves side by side where the budget
// covers them, else one after the other: how many runs are in flight decides), attaches the result; 0, or an error code with *err
extern "C" int stdin_resolve_deferred(const zksp_client* c, zksp_stdin* s, std::string* err, unsigned budget);
struct zksp_proof {
  std::vector<uint8_t> bytes;
#ifdef ZKSP_COMPONENT
  zksp::ProofHeader hdr;  // (a keccak-chip component proof, format v2)
#endif
  zksp::MachineHeader mhdr;
  uint32_t version = 0;
};

// api_machine.cpp: the proof object (format mach::kMachineVersion) from an execution record, the chip heights, the aggregation leaves and a fetched body
int machine_proof_from_parts(const zksp_pk* pk, const zksp::ExecutionRecord& r, const int* log_heights, const uint32_t* handover_pc /* [kNumCpuInst - 1] */,
                             const std::vector<uint32_t>& agg_leaves, const std::vector<uint32_t>& agg_keys,
                             const zksp::LeafCheckLog* leaf_check, const uint32_t* body, size_t body_words, zksp_proof** out);
