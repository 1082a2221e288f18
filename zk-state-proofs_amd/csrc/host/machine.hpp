// The RISC-V machine the multi-chip proof is about: program and
// memory-image tables derived from the guest ELF, and the traced execution that feeds every chip.
//
// Replaces, for this repository's own proof system, what sp1-core-executor's ExecutionRecord and
// sp1-core-machine's Program / MemoryProgram tables are to SP1 (reference Cargo.lock:7096, :7130;
// reached only beneath `client.prove(&pk, stdin).run()`, prover/src/bin/main.rs:71-74).  The
// statement a machine proof establishes is the reference's: the committed guest
// (circuits/sp1-merkle-proof/src/main.rs:4-14) ran crypto_ops::verify_merkle_proof
// (crypto-ops/src/lib.rs:8-23) to HALT(0) and committed these public values.
//
// Record layouts are plain arrays of u32/u64 because the same bytes go to the device (trace
// expansion kernels) and, in tests, to the CPU oracle.
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "executor.hpp"

namespace zksp {

// AIR opcodes (one selector column each in the CPU chip; 0 = not provable)
enum AirOp : uint32_t {
  AIR_NONE = 0,
  AIR_ADD, AIR_SUB, AIR_XOR, AIR_OR, AIR_AND, AIR_SLL, AIR_SRL, AIR_SRA, AIR_SLT, AIR_SLTU,
  AIR_JAL, AIR_JALR, AIR_BEQ, AIR_BNE, AIR_BLT, AIR_BGE, AIR_BLTU, AIR_BGEU,
  AIR_LB, AIR_LH, AIR_LW, AIR_LBU, AIR_LHU, AIR_SB, AIR_SH, AIR_SW,
  AIR_MUL, AIR_MULHU, AIR_ECALL, AIR_KECCAK,
  AIR_MULH, AIR_MULHSU, AIR_DIV, AIR_DIVU, AIR_REM, AIR_REMU,
  AIR_NUM_OPS  // 37: ops are 1..36
};

// One row of the preprocessed Program table (9 u32).  The table ends with the padding instruction: `jal x0, 0` at
// the first address after the text segment, which the CPU rows after HALT execute (it jumps to itself).
struct ProgramRow {
  uint32_t pc, op, wr, use2, rd, rs1, rs2, imm, tgt;
};
// One row of the preprocessed memory-image table: registers x0..x31 at addresses 0..31 (value 0),
// then every word of every PT_LOAD segment (p_memsz extent, zero beyond p_filesz).
struct ImageRow {
  uint32_t addr, val;
};

struct MachineProgram {
  std::vector<ProgramRow> rows;  // one per instruction word of the text segment, in address order, then the padding row
  uint32_t pad_pc() const { return rows.empty() ? 0 : rows.back().pc; }
  std::vector<ImageRow> image;   // sorted by addr
  uint32_t entry = 0;
  uint32_t text_base = 0;
  int keccak_mode = 2;           // KeccakMode: kReplace patches keccakf entry points to AIR_KECCAK
  int log_prog = 0, log_image = 0;  // table heights (powers of two covering rows / image)
};

// One executed cycle (12 u32).  Everything else a CPU-chip row holds follows from the Program
// table row of `pc` and the row index (ts = 4 * (index + 1)).
struct CycleRec {
  uint32_t pc;
  uint32_t a;       // value written to rd (or the op's result when rd = x0)
  uint32_t b;       // reg[rs1]
  uint32_t c;       // reg[rs2], or the immediate when the instruction has one
  uint32_t m;       // memory slot: word read (loads, stores: the old word; ecall: x11)
  uint32_t mv;      // memory slot: word left behind
  uint32_t w_prev;  // previous value of rd
  uint32_t r1_pts, r2_pts, m_pts, w_pts;  // previous access time of each slot's address
  uint32_t pad;
};
// One keccak precompile call: 25 u64 in, previous access times of its 50 words.
struct KeccakCall {
  uint32_t ts, ptr;
  uint64_t in[25];
  uint32_t pts[50];
};
struct MemFinalRec {
  uint32_t addr, init, fin, fin_ts, is_init;  // is_init: 0 an image word, 1 a hinted word (a HINT_READ covers it), 2 any other: starts as zero
};
struct MulRec {
  uint32_t hi, b, c;  // hi: 0 mul, 1 mulhu, 2 mulh, 3 mulhsu (also: the |q| * |d| products the divider chip asks for, as 0 and 1)
};

// Page-backed storage for the per-cycle records (19 MB and more per run): mapped and unmapped directly.  With
// the default allocator such blocks end up in malloc's arenas (its mmap threshold adapts upwards after the first
// free), and returning hundreds of megabytes from there stalls whichever thread frees next.
template <class T>
struct PageAllocator {
  using value_type = T;
  PageAllocator() = default;
  template <class U>
  PageAllocator(const PageAllocator<U>&) {}
  T* allocate(size_t n);
  void deallocate(T* p, size_t n) noexcept;
  template <class U>
  bool operator==(const PageAllocator<U>&) const { return true; }
  template <class U>
  bool operator!=(const PageAllocator<U>&) const { return false; }
};
void* page_alloc(size_t bytes);              // throws std::bad_alloc
void page_free(void* p, size_t bytes) noexcept;
template <class T>
T* PageAllocator<T>::allocate(size_t n) { return static_cast<T*>(page_alloc(n * sizeof(T))); }
template <class T>
void PageAllocator<T>::deallocate(T* p, size_t n) noexcept { page_free(p, n * sizeof(T)); }

// What verifying a leaf proof leaves behind for a proof ABOUT that verification (stage 2b; the
// reference's circuits/sp1-merkle-proof-recursive/src/main.rs:3-5 is a todo!()): every Poseidon2 permutation of the query
// phase as a row record of the Poseidon2 chip (32 words each: the sponges over the opened rows with their Horner sums, the
// Merkle paths with their injections, the FRI leaves and paths), every duplex of the leaf's Fiat-Shamir transcript as a row
// record of the transcript chip (32 words), the 31 rows per query of the query chip (the canonical bits of the query's index
// word, the folding chain, the reduced openings), and the public bus tuples (16 words each) that state WHAT was checked: the
// transcript's blocks (header, commitment roots, cumulative sums, the root of the opened values, FRI roots, final constant,
// witness), the preprocessed root, one tuple of constants per leaf and one per height (which depend on the leaf's challenges,
// but on nothing the query phase opens), the proof-of-work word.  All canonical words.
// (storage whose resize() leaves new words unwritten: the 11 MB of row records per leaf are written by several threads, which
// is also where their pages are first touched)
template <class T>
struct RawAllocator : std::allocator<T> {
  template <class U>
  struct rebind { using other = RawAllocator<U>; };
  template <class U>
  void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
  template <class U, class... A>
  void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
using RowWords = std::vector<uint32_t, RawAllocator<uint32_t>>;
struct LeafCheckLog {
  RowWords p2_rows, qr_rows;
  std::vector<uint32_t> tr_rows, pub_tuples;
  // several leaf proofs checked beside one run: the k-th leaf's tags, root ids and tuples carry the leaf index k
  uint32_t leaf_index = 0, n_leaves = 0;
  void append(const LeafCheckLog& o) {
    const LeafCheckLog* one[1] = {&o};
    append_all(one, 1);
  }
  // the logs of several leaves, in order; the big record lists are copied side by side
  void append_all(const LeafCheckLog* const* parts, size_t n);
};

struct MachineTrace {
  ExecutionRecord rec;  // cycles, exit code, public values, digests, error text
  std::vector<CycleRec, PageAllocator<CycleRec>> cycles;
  std::vector<KeccakCall> keccak;
  std::vector<MemFinalRec> memfinal;  // every image address and every other touched address, strictly increasing
  std::vector<MulRec> muls;
  std::vector<uint32_t> prog_mult;    // per Program row (the padding row: 0 here; its fetches depend on the chip heights)
  std::vector<uint32_t> alu_idx;      // cycles that occupy a row of the ALU chip (sll srl sra slt, blt bge), in order
  std::vector<uint32_t> bw_idx;       // cycles that occupy a row of the bitwise chip (xor or and), in order
  std::vector<uint32_t> sub_idx;      // cycles that occupy a row of the sub-word chip (lb lh lbu lhu sb sh), in order
  std::vector<uint32_t> ecall_idx;    // the ecall cycles (one row of the ecall chip each), in order
  std::vector<uint32_t> div_idx;      // cycles that occupy a row of the divider chip (div divu rem remu), in order
  uint32_t x0_last = 0;               // last access time of x0 by a real cycle (the first padding row consumes it)
  size_t hint_words = 0;              // words covered by the run's HINT_READs: the rows of the hint chip
  std::vector<uint32_t> agg_leaves;   // aggregation payload (row f4): 8 canonical words per supplied digest, or none
  std::vector<uint32_t> agg_keys;     // their heap keys (empty: n + j, the leaves of a full tree of n = a power of two)
  size_t agg_rows = 0;                // node rows of the Poseidon2 chip: the ancestors of the supplied keys (set with the payload)
  std::shared_ptr<const LeafCheckLog> leaf_check;  // leaf-proof check to prove beside the run (row f4, stage 2a), or none
  size_t p2_rows() const { return agg_rows + (leaf_check ? leaf_check->p2_rows.size() / 32 : 0); }
  size_t qr_rows() const { return leaf_check ? leaf_check->qr_rows.size() / 132 : 0; }
  size_t tr_rows() const { return leaf_check ? leaf_check->tr_rows.size() / 32 : 0; }
};

// How many rows of each event-sized chip a run needs; a batch is proven with the heights of the element-wise maximum.
struct MachineCounts {
  size_t cycles = 0, alu = 0, sub = 0, bw = 0, keccak = 0, memfinal = 0, muls = 0, agg = 0 /* Poseidon2 chip rows */, ecall = 0, fold = 0 /* query chip rows */, div = 0, tr = 0 /* transcript chip rows */, hint = 0 /* hint chip rows */;
  void cover(const MachineTrace& t) {
    cycles = std::max(cycles, t.cycles.size()); alu = std::max(alu, t.alu_idx.size()); sub = std::max(sub, t.sub_idx.size());
    bw = std::max(bw, t.bw_idx.size()); agg = std::max(agg, t.p2_rows()); ecall = std::max(ecall, t.ecall_idx.size()); fold = std::max(fold, t.qr_rows()); tr = std::max(tr, t.tr_rows()); hint = std::max(hint, t.hint_words); div = std::max(div, t.div_idx.size());
    keccak = std::max(keccak, t.keccak.size()); memfinal = std::max(memfinal, t.memfinal.size()); muls = std::max(muls, t.muls.size());
  }
  void cover(const MachineCounts& o) {
    cycles = std::max(cycles, o.cycles); alu = std::max(alu, o.alu); sub = std::max(sub, o.sub); bw = std::max(bw, o.bw);
    agg = std::max(agg, o.agg); ecall = std::max(ecall, o.ecall); keccak = std::max(keccak, o.keccak); fold = std::max(fold, o.fold); tr = std::max(tr, o.tr); hint = std::max(hint, o.hint); div = std::max(div, o.div);
    memfinal = std::max(memfinal, o.memfinal); muls = std::max(muls, o.muls);
  }
};

// rows of the first of two instances of a chip: the largest power of two strictly below the count (at least 32);
// the second instance takes the rest, rounded up to a power of two (at least 32)
inline size_t split_rows(size_t n) {
  size_t h0 = 32;
  while (2 * h0 < n) h0 *= 2;
  return h0;
}
inline size_t split_rest_rows(size_t n) {
  const size_t h0 = split_rows(n);
  size_t h1 = 32;
  while (h1 < (n > h0 ? n - h0 : 1)) h1 *= 2;
  return h1;
}

// Builds the two preprocessed tables.  Returns "" or an error.
std::string build_machine_program(const ElfImage& elf, KeccakMode mode, MachineProgram* out);

// Executes the guest and records what every chip needs.  rec.error is set (and the trace is
// unusable) on executor faults and on instructions the AIR does not cover.
void trace_execute(const ElfImage& elf, const MachineProgram& prog, const std::vector<std::vector<uint8_t>>& stdin_entries,
                   uint64_t max_cycles, MachineTrace* out);

}  // namespace zksp
