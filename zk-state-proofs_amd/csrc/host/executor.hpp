// RV32IM executor for the reference's committed SP1 guest
// (reference circuits/elf/riscv32im-succinct-zkvm-elf, built from
// circuits/sp1-merkle-proof/src/main.rs:4-14).  This is hot-path row a2 of
// the witness producer.  It stays on the host (sequential by
// nature) and feeds the device prover with keccak-f[1600] permutation events.
//
// Guest-side ABI (SURVEY.md section 8b "Guest-side ABI", decoded from the ELF):
//   ecall code in t0; args a0,a1,a2
//   0xf0 HINT_LEN   -> result in t0
//   0xf1 HINT_READ  (a0=ptr, a1=len)
//   0x02 WRITE      (a0=fd, a1=ptr, a2=len)  fd 3 = public values, 1/2 = stdout/stderr
//   0x10 COMMIT     (a0=word index, a1=digest word)
//   0x1a COMMIT_DEFERRED_PROOFS (a0=word index, a1=word)
//   0x00 HALT       (a0=exit code)
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace zksp {

struct ElfImage {
  uint32_t entry = 0;
  uint32_t text_base = 0;             // first executable byte
  std::vector<uint32_t> text;         // instruction words of the RX segment
  struct Seg { uint32_t vaddr; std::vector<uint8_t> bytes; uint32_t memsz; };
  std::vector<Seg> segs;              // all PT_LOAD segments (incl. text)
  std::vector<uint32_t> keccakf_entries;  // FUNC symbols whose name contains "keccakf"
  uint32_t max_addr = 0;
  std::array<uint8_t, 32> sha256{};   // digest of the file bytes
  // Decoded once by load_elf (text.size() + 1 entries, the last a sentinel that faults when
  // execution runs off the end): {op | 0x80 at keccakf entry points, rd (0 -> 32, a sink
  // register), rs1, rs2, imm}.  Decoding 39 k words per run was a quarter of a run.
  struct Insn { uint8_t op, rd, rs1, rs2; int32_t imm; };
  std::vector<Insn> code;
};

// Parses an ELF32 little-endian RISC-V executable. Returns "" on success.
std::string load_elf(const uint8_t* data, size_t len, ElfImage* out);

enum class KeccakMode : int {
  kSoftware = 0,  // run tiny-keccak's keccakf as RISC-V code; record nothing
  kObserve = 1,   // run it as RISC-V code, record (state_in) at every entry
  kReplace = 2,   // record state_in, apply keccak-f natively, return (precompile shape)
};

struct KeccakEvent {
  uint64_t state_in[25];
  uint32_t state_ptr;
  uint64_t cycle;
};

struct ExecutionRecord {
  uint64_t cycles = 0;
  uint32_t exit_code = 0;
  bool halted = false;
  std::string error;                  // executor-level fault (not a guest panic)
  std::vector<uint8_t> public_values; // bytes written to fd 3
  std::array<uint32_t, 8> pv_digest{};        // COMMIT words
  std::array<uint32_t, 8> deferred_digest{};  // COMMIT_DEFERRED_PROOFS words
  std::string stdout_text, stderr_text;
  std::vector<KeccakEvent> keccak_events;
  uint64_t memory_ops = 0;            // lw/lh/lb/lbu/lhu/sw/sh/sb executed
  // traced runs only: loads (and keccak-state reads) of bytes that were neither in the program image, nor hinted, nor
  // written before.  Memory outside the image starts with prover-chosen contents in the proof (as hinted input must); a run
  // that never reads such a byte cannot be steered by them.  The committed guest never does (tests/test_machine.py).
  uint64_t uninit_reads = 0;
  // traced with ZKSP_UNINIT_FILL (fresh memory filled with a chosen byte: an analysis aid of tests/test_machine.py): such a
  // trace is for inspection only, the provers refuse it
  bool analysis_fill = false;
  uint64_t syscall_counts[256] = {0}; // indexed by low byte of the code
  std::vector<uint64_t> opcode_hist;  // indexed by Op (filled when want_hist)
  // a run that ended on a jump outside the text: where to (a function called through ExecOptions::call_pc returns to the
  // address behind the text, which is how its return is told from a fault), and the 200 bytes at call_a0 when it ended
  uint32_t fault_target = 0;
  uint64_t call_state_out[25] = {0};
};

struct ExecOptions {
  KeccakMode keccak_mode = KeccakMode::kObserve;
  uint64_t max_cycles = 1ull << 28;
  bool want_hist = false;
  // call ONE function of the guest instead of running it from its entry point: pc = call_pc, a0 = call_a0 pointing at
  // the 200 bytes of call_state, sp = call_sp, ra = the address behind the text (key generation checks with this that a
  // function the precompile shape replaces by the keccak chip computes keccak-f: check_keccakf_entries)
  uint32_t call_pc = 0, call_a0 = 0, call_sp = 0;
  const uint64_t* call_state = nullptr;
};

// Opcode ids (also indexes ExecutionRecord::opcode_hist).
enum Op : uint8_t {
  OP_INVALID = 0,
  OP_LUI, OP_AUIPC, OP_JAL, OP_JALR,
  OP_BEQ, OP_BNE, OP_BLT, OP_BGE, OP_BLTU, OP_BGEU,
  OP_LB, OP_LH, OP_LW, OP_LBU, OP_LHU,
  OP_SB, OP_SH, OP_SW,
  OP_ADDI, OP_SLTI, OP_SLTIU, OP_XORI, OP_ORI, OP_ANDI, OP_SLLI, OP_SRLI, OP_SRAI,
  OP_ADD, OP_SUB, OP_SLL, OP_SLT, OP_SLTU, OP_XOR, OP_SRL, OP_SRA, OP_OR, OP_AND,
  OP_MUL, OP_MULH, OP_MULHSU, OP_MULHU, OP_DIV, OP_DIVU, OP_REM, OP_REMU,
  OP_ECALL, OP_FENCE, OP_UNIMP,
  OP_COUNT
};
const char* op_name(int op);

// Runs the guest on one or more stdin entries (each entry is one
// SP1Stdin::write buffer, already framed by the caller).
ExecutionRecord execute(const ElfImage& elf, const std::vector<std::vector<uint8_t>>& stdin_entries,
                        const ExecOptions& opt);
// Every function the ELF names keccakf, run on test states by the executor (software, nothing replaced) and compared with
// keccak-f[1600]: "" if all agree, else what differs.  The precompile shape's verifying key stands for "calls of these
// addresses are keccak-f"; this is the check, at key generation, that they are - on vectors, not a proof.
std::string check_keccakf_entries(const ElfImage& elf);

// keccak-f[1600] on 25 little-endian lanes (state[x + 5*y]).
void keccak_f1600(uint64_t st[25]);
// keccak-256 (0x01 padding), used for vk hashing and host-side checks.
void keccak256(const uint8_t* data, size_t len, uint8_t out[32]);
void sha256(const uint8_t* data, size_t len, uint8_t out[32]);

}  // namespace zksp
