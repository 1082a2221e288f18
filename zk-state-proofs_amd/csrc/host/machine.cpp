// See machine.hpp.  The traced interpreter is the plain, one-switch form of executor.cpp's
// threaded run loop (same faults, same cycle accounting: tests/test_machine.py compares them on
// every golden fixture) plus the bookkeeping of the offline memory argument: the previous access
// time of every register and memory word, the set of touched addresses with their first and last
// values, multiplicities of the Program and Image tables.
#include "machine.hpp"

#include <sys/mman.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <utility>
#include <thread>
#include <vector>

namespace zksp {

// Mapped regions are recycled through a small pool (at most kPagePoolLimit bytes, exact sizes): a region that comes
// back from the pool is already faulted in, so tracing the next run does not pay 5 000 page faults again, and
// giving a region back is a push instead of an munmap.
namespace {
constexpr size_t kPagePoolLimit = (size_t)1 << 30;
std::mutex g_page_mu;
std::vector<std::pair<void*, size_t>> g_page_pool;
size_t g_page_pool_bytes = 0;
size_t page_round(size_t bytes) { return ((bytes ? bytes : 1) + 4095) & ~(size_t)4095; }
}  // namespace

void* page_alloc(size_t bytes) {
  const size_t sz = page_round(bytes);
  {
    std::lock_guard<std::mutex> lk(g_page_mu);
    for (size_t i = g_page_pool.size(); i-- > 0;)
      if (g_page_pool[i].second == sz) {
        void* p = g_page_pool[i].first;
        g_page_pool[i] = g_page_pool.back();
        g_page_pool.pop_back();
        g_page_pool_bytes -= sz;
        return p;
      }
  }
  void* p = mmap(nullptr, sz, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) throw std::bad_alloc();
  return p;
}
void page_free(void* p, size_t bytes) noexcept {
  if (!p) return;
  const size_t sz = page_round(bytes);
  {
    std::lock_guard<std::mutex> lk(g_page_mu);
    if (g_page_pool_bytes + sz <= kPagePoolLimit) {
      try {
        g_page_pool.emplace_back(p, sz);
        g_page_pool_bytes += sz;
        return;
      } catch (...) {
      }
    }
  }
  (void)munmap(p, sz);
}

namespace {

constexpr uint64_t kMemBytes = 0x78000000ull;
// guest accesses below kRegSpace are refused (addresses 0..31 name registers on the memory bus) and so are accesses from
// kDataTop up: the AIR looks an address's high limb up in 1 .. 0x77FE (air_machine.hpp kAddrHiMax), so they have no proof
constexpr uint32_t kRegSpace = 0x10000, kDataTop = 0x77FF0000u;

struct Map {
  uint8_t* base = nullptr;
  uint64_t len;
  explicit Map(uint64_t n) : len(n) {
    void* p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    base = (p == MAP_FAILED) ? nullptr : (uint8_t*)p;
  }
  ~Map() {
    if (base) munmap(base, len);
  }
};

struct SegRange {
  uint32_t lo, hi, row0;  // byte range [lo, hi), first image row
};

}  // namespace

std::string build_machine_program(const ElfImage& elf, KeccakMode mode, MachineProgram* out) {
  *out = MachineProgram();
  out->entry = elf.entry;
  out->text_base = elf.text_base;
  out->keccak_mode = (int)mode;
  if (elf.code.size() != elf.text.size() + 1) return "ELF image was not decoded";
  const size_t n = elf.text.size();
  out->rows.resize(n);
  for (size_t i = 0; i < n; ++i) {
    const ElfImage::Insn& in = elf.code[i];
    const uint32_t pc = elf.text_base + 4 * (uint32_t)i;
    const uint8_t op = in.op & 0x7f;
    const bool hook = (in.op & 0x80) != 0;
    ProgramRow r{pc, AIR_NONE, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t rd = in.rd == 32 ? 0 : in.rd, imm = (uint32_t)in.imm;
    auto alu = [&](uint32_t a, bool has_imm) {
      r.op = a; r.rd = rd; r.wr = rd != 0; r.rs1 = in.rs1;
      if (has_imm) { r.use2 = 0; r.imm = imm; } else { r.use2 = 1; r.rs2 = in.rs2; }
    };
    auto branch = [&](uint32_t a) { r.op = a; r.rs1 = in.rs1; r.rs2 = in.rs2; r.use2 = 1; r.tgt = pc + imm; };
    auto load = [&](uint32_t a) { r.op = a; r.rd = rd; r.wr = rd != 0; r.rs1 = in.rs1; r.imm = imm; };
    auto store = [&](uint32_t a) { r.op = a; r.rs1 = in.rs1; r.rs2 = in.rs2; r.use2 = 1; r.imm = imm; };
    if (hook && mode == KeccakMode::kReplace) {
      // the keccakf entry point as the precompile call the `sp1` feature of crypto-ops would compile to
      // (reference crypto-ops/src/keccak.rs:1-4): permute the 25 lanes at a0 in place, return to ra
      r.op = AIR_KECCAK; r.rs1 = 1; r.rs2 = 10; r.use2 = 1;
    } else {
      switch (op) {
        case OP_LUI: r.op = AIR_ADD; r.rd = rd; r.wr = rd != 0; r.imm = imm; break;
        case OP_AUIPC: r.op = AIR_ADD; r.rd = rd; r.wr = rd != 0; r.imm = pc + imm; break;
        case OP_JAL: r.op = AIR_JAL; r.rd = rd; r.wr = rd != 0; r.imm = pc + 4; r.tgt = pc + imm; break;
        case OP_JALR: r.op = AIR_JALR; r.rd = rd; r.wr = rd != 0; r.rs1 = in.rs1; r.imm = imm; r.tgt = pc + 4; break;
        case OP_BEQ: branch(AIR_BEQ); break;
        case OP_BNE: branch(AIR_BNE); break;
        case OP_BLT: branch(AIR_BLT); break;
        case OP_BGE: branch(AIR_BGE); break;
        case OP_BLTU: branch(AIR_BLTU); break;
        case OP_BGEU: branch(AIR_BGEU); break;
        case OP_LB: load(AIR_LB); break;
        case OP_LH: load(AIR_LH); break;
        case OP_LW: load(AIR_LW); break;
        case OP_LBU: load(AIR_LBU); break;
        case OP_LHU: load(AIR_LHU); break;
        case OP_SB: store(AIR_SB); break;
        case OP_SH: store(AIR_SH); break;
        case OP_SW: store(AIR_SW); break;
        case OP_ADDI: alu(AIR_ADD, true); break;
        case OP_SLTI: alu(AIR_SLT, true); break;
        case OP_SLTIU: alu(AIR_SLTU, true); break;
        case OP_XORI: alu(AIR_XOR, true); break;
        case OP_ORI: alu(AIR_OR, true); break;
        case OP_ANDI: alu(AIR_AND, true); break;
        case OP_SLLI: alu(AIR_SLL, true); break;
        case OP_SRLI: alu(AIR_SRL, true); break;
        case OP_SRAI: alu(AIR_SRA, true); break;
        case OP_ADD: alu(AIR_ADD, false); break;
        case OP_SUB: alu(AIR_SUB, false); break;
        case OP_SLL: alu(AIR_SLL, false); break;
        case OP_SLT: alu(AIR_SLT, false); break;
        case OP_SLTU: alu(AIR_SLTU, false); break;
        case OP_XOR: alu(AIR_XOR, false); break;
        case OP_SRL: alu(AIR_SRL, false); break;
        case OP_SRA: alu(AIR_SRA, false); break;
        case OP_OR: alu(AIR_OR, false); break;
        case OP_AND: alu(AIR_AND, false); break;
        case OP_MUL: alu(AIR_MUL, false); break;
        case OP_MULHU: alu(AIR_MULHU, false); break;
        case OP_MULH: alu(AIR_MULH, false); break;
        case OP_MULHSU: alu(AIR_MULHSU, false); break;
        case OP_DIV: alu(AIR_DIV, false); break;
        case OP_DIVU: alu(AIR_DIVU, false); break;
        case OP_REM: alu(AIR_REM, false); break;
        case OP_REMU: alu(AIR_REMU, false); break;
        case OP_ECALL: r.op = AIR_ECALL; r.rd = 5; r.wr = 1; r.rs1 = 5; r.rs2 = 10; r.use2 = 1; break;
        case OP_FENCE: r.op = AIR_ADD; break;  // no architectural effect: x0 = x0 + 0, not written
        default: break;  // unimp, invalid: no AIR row can match
      }
    }
    out->rows[i] = r;
  }
  {
    // the padding instruction: jal x0, 0 right after the text segment
    const uint32_t pc = elf.text_base + 4 * (uint32_t)n;
    out->rows.push_back(ProgramRow{pc, AIR_JAL, 0, 0, 0, 0, 0, 0, pc});
  }
  for (uint32_t i = 0; i < 32; ++i) out->image.push_back({i, 0});
  std::vector<const ElfImage::Seg*> segs;
  for (const auto& s : elf.segs) segs.push_back(&s);
  std::sort(segs.begin(), segs.end(), [](const ElfImage::Seg* a, const ElfImage::Seg* b) { return a->vaddr < b->vaddr; });
  uint32_t prev_end = kRegSpace;
  for (const ElfImage::Seg* s : segs) {
    const uint64_t extent = std::max<uint64_t>(s->bytes.size(), s->memsz);
    if (extent == 0) continue;
    if (s->vaddr < prev_end) return "segments overlap or start inside the register space";
    if ((uint64_t)s->vaddr + extent > kMemBytes) return "segment beyond guest memory";
    const uint32_t end = (uint32_t)((s->vaddr + extent + 3) & ~3ull);
    for (uint32_t a = s->vaddr; a < end; a += 4) {
      uint32_t v = 0;
      for (int k = 0; k < 4; ++k) {
        const uint64_t off = (uint64_t)(a - s->vaddr) + k;
        if (off < s->bytes.size()) v |= (uint32_t)s->bytes[off] << (8 * k);
      }
      out->image.push_back({a, v});
    }
    prev_end = end;
  }
  auto clog2 = [](size_t v) { int l = 0; while (((size_t)1 << l) < v) ++l; return l; };
  out->log_prog = std::max(5, clog2(out->rows.size()));
  out->log_image = std::max(5, clog2(out->image.size()));
  return "";
}

void trace_execute(const ElfImage& elf, const MachineProgram& prog, const std::vector<std::vector<uint8_t>>& stdin_entries,
                   uint64_t max_cycles, MachineTrace* out) {
  *out = MachineTrace();
  ExecutionRecord& rec = out->rec;
  Map mem(kMemBytes), shadow(kMemBytes);  // shadow: one u32 per word = 1 + index into `touched`
  if (!mem.base || !shadow.base) { rec.error = "mmap of guest memory failed"; return; }
  uint8_t* M = mem.base;
  uint32_t* SH = reinterpret_cast<uint32_t*>(shadow.base);
  std::vector<SegRange> ranges;
  {
    // image rows after the 32 registers are the segments' words in address order
    size_t row = 32;
    while (row < prog.image.size()) {
      size_t e = row;
      while (e + 1 < prog.image.size() && prog.image[e + 1].addr == prog.image[e].addr + 4) ++e;
      ranges.push_back({prog.image[row].addr, prog.image[e].addr + 4, (uint32_t)row});
      row = e + 1;
    }
    for (size_t r = 32; r < prog.image.size(); ++r) memcpy(M + prog.image[r].addr, &prog.image[r].val, 4);
  }
  auto image_row = [&](uint32_t w) -> int64_t {
    for (const SegRange& s : ranges)
      if (w >= s.lo && w < s.hi) return (int64_t)s.row0 + ((w - s.lo) >> 2);
    return -1;
  };
  struct Touched { uint32_t addr, init, last_ts, is_init, valid /* bytes that are image, hinted or written */; };
  std::vector<Touched> touched;
  std::vector<std::pair<uint32_t, uint32_t>> hinted;  // [lo, hi) of every HINT_READ
  // ZKSP_UNINIT_FILL=<byte>: fresh memory (not image, not hinted) starts filled with that byte instead of zero - in the
  // proof its contents are the prover's choice, and tests/test_machine.py uses this to show that the guest's public values
  // do not depend on them
  const int uninit_fill = getenv("ZKSP_UNINIT_FILL") ? (int)(strtoul(getenv("ZKSP_UNINIT_FILL"), nullptr, 0) & 0xff) : -1;
  rec.analysis_fill = uninit_fill >= 0;
  out->cycles.reserve((size_t)1 << 19);  // virtual pages only: what is not written is never touched
  out->prog_mult.assign(prog.rows.size(), 0);
  uint32_t x[32] = {0}, reg_ts[32] = {0};
  const uint32_t text_lo = prog.text_base, text_bytes = 4 * (uint32_t)(prog.rows.size() - 1);  // the padding row is not code
  size_t stdin_pos = 0;
  uint64_t cycles = 0;

#define FAULT(msg) do { rec.error = (msg); goto done; } while (0)
#define CHECK_ADDR(ad, n) if ((uint64_t)(ad) + (n) > kDataTop) FAULT("memory access out of range")

  // previous access time of word w, which is touched at time `now`
  // `reads` / `writes`: masks of the word's bytes the access reads and writes (uninitialised-read accounting)
  auto touch = [&](uint32_t w, uint32_t now, uint32_t reads, uint32_t writes) -> uint32_t {
    uint32_t id = SH[w >> 2];
    if (id == 0) {
      uint32_t v;
      memcpy(&v, M + w, 4);
      const int64_t ir = image_row(w);
      bool known = ir >= 0;
      for (const auto& hr : hinted) known = known || (w >= hr.first && w < hr.second);
      if (!known && uninit_fill >= 0) {  // analysis aid: what a prover may choose as this word's initial contents
        memset(M + w, uninit_fill, 4);
        memcpy(&v, M + w, 4);
      }
      // (is_init: 0 an image word, 1 a hinted word - its initial value is the input's -, 2 any other address: it starts as zero)
      touched.push_back({w, v, 0, ir >= 0 ? 0u : known ? 1u : 2u, known ? 0xfu : 0u});
      id = (uint32_t)touched.size();
      SH[w >> 2] = id;
    }
    Touched& t = touched[id - 1];
    if ((t.valid & reads) != reads) ++rec.uninit_reads;
    t.valid |= writes;
    const uint32_t prev = t.last_ts;
    t.last_ts = now;
    return prev;
  };

  {
    uint32_t pc = prog.entry;
    for (;;) {
      const uint32_t off = pc - text_lo;
      if (off >= text_bytes || (pc & 3)) FAULT("pc outside text segment");
      const size_t idx = off >> 2;
      const ProgramRow& r = prog.rows[idx];
      if (cycles >= max_cycles) FAULT("cycle limit exceeded");
      if (r.op == AIR_NONE) {
        const uint8_t eop = elf.code[idx].op & 0x7f;
        if (eop == OP_UNIMP) FAULT("unimp executed");
        if (eop == OP_INVALID) FAULT("illegal instruction");
        FAULT(std::string("instruction not covered by the machine AIR: ") + op_name(eop));
      }
      const uint32_t ts = 4 * (uint32_t)(cycles + 1);
      ++cycles;
      out->prog_mult[idx]++;
      CycleRec c{};
      c.pc = pc;
      c.b = x[r.rs1];
      c.r1_pts = reg_ts[r.rs1]; reg_ts[r.rs1] = ts;
      if (r.use2) {
        c.c = x[r.rs2];
        c.r2_pts = reg_ts[r.rs2]; reg_ts[r.rs2] = ts + 1;
      } else {
        c.c = r.imm;
      }
      const uint32_t a = c.b, b = c.c;
      uint32_t res = 0, next = pc + 4;
      bool halt = false, alu_event = false, bw_event = false;
      switch (r.op) {
        case AIR_ADD: res = a + b; break;
        case AIR_SUB: res = a - b; break;
        case AIR_XOR: res = a ^ b; bw_event = true; break;
        case AIR_OR: res = a | b; bw_event = true; break;
        case AIR_AND: res = a & b; bw_event = true; break;
        case AIR_SLL: res = a << (b & 31); alu_event = true; break;
        case AIR_SRL: res = a >> (b & 31); alu_event = true; break;
        case AIR_SRA: res = (uint32_t)((int32_t)a >> (b & 31)); alu_event = true; break;
        case AIR_SLT: res = (int32_t)a < (int32_t)b; alu_event = true; break;
        case AIR_SLTU: res = a < b; break;  // an unsigned comparison: the CPU row does it itself
        case AIR_JAL: res = r.imm; next = r.tgt; break;
        case AIR_JALR: res = r.tgt; next = (a + r.imm) & ~1u; break;
        case AIR_BEQ: if (a == b) next = r.tgt; break;
        case AIR_BNE: if (a != b) next = r.tgt; break;
        case AIR_BLT: if ((int32_t)a < (int32_t)b) next = r.tgt; alu_event = true; break;
        case AIR_BGE: if ((int32_t)a >= (int32_t)b) next = r.tgt; alu_event = true; break;
        case AIR_BLTU: if (a < b) next = r.tgt; break;
        case AIR_BGEU: if (a >= b) next = r.tgt; break;
        case AIR_MUL: res = a * b; out->muls.push_back({0, a, b}); break;
        case AIR_MULHU: res = (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); out->muls.push_back({1, a, b}); break;
        case AIR_MULH: res = (uint32_t)(((int64_t)(int32_t)a * (int64_t)(int32_t)b) >> 32); out->muls.push_back({2, a, b}); break;
        case AIR_MULHSU: res = (uint32_t)(((int64_t)(int32_t)a * (int64_t)(uint64_t)b) >> 32); out->muls.push_back({3, a, b}); break;
        case AIR_DIV: case AIR_DIVU: case AIR_REM: case AIR_REMU: {
          // RISC-V: x / 0 = all ones, x % 0 = x; -2^31 / -1 = -2^31, remainder 0
          const bool sg = r.op == AIR_DIV || r.op == AIR_REM;
          uint32_t q, rm;
          if (b == 0) { q = 0xffffffffu; rm = a; }
          else if (sg && a == 0x80000000u && b == 0xffffffffu) { q = a; rm = 0; }
          else if (sg) { q = (uint32_t)((int32_t)a / (int32_t)b); rm = (uint32_t)((int32_t)a % (int32_t)b); }
          else { q = a / b; rm = a % b; }
          res = (r.op == AIR_DIV || r.op == AIR_DIVU) ? q : rm;
          out->div_idx.push_back((uint32_t)(cycles - 1));
          if (b != 0) {  // the divider chip asks the multiplier chip for |q| * |d| (low word, and the high word: zero)
            const uint32_t aq = sg && (int32_t)q < 0 && !(a == 0x80000000u && b == 0xffffffffu) ? 0u - q : q;
            const uint32_t ad = sg && (int32_t)b < 0 ? 0u - b : b;
            out->muls.push_back({0, aq, ad});
            out->muls.push_back({1, aq, ad});
          }
          break;
        }
        case AIR_LB: case AIR_LH: case AIR_LW: case AIR_LBU: case AIR_LHU: {
          const uint32_t ad = a + r.imm;
          const int sz = (r.op == AIR_LW) ? 4 : (r.op == AIR_LH || r.op == AIR_LHU) ? 2 : 1;
          if (ad & (sz - 1)) FAULT(sz == 4 ? "unaligned lw" : (r.op == AIR_LH ? "unaligned lh" : "unaligned lhu"));
          CHECK_ADDR(ad, sz);
          if (ad < kRegSpace) FAULT("guest access below 0x10000 (register-mapped addresses)");
          const uint32_t w = ad & ~3u;
          c.m_pts = touch(w, ts + 1, ((1u << sz) - 1) << (ad & 3), 0);  // a load reads its word as the row's second access
          memcpy(&c.m, M + w, 4);
          c.mv = c.m;
          const uint32_t sh = 8 * (ad & 3);
          if (r.op == AIR_LW) res = c.m;
          else if (r.op == AIR_LHU) res = (c.m >> sh) & 0xffff;
          else if (r.op == AIR_LH) res = (uint32_t)(int32_t)(int16_t)((c.m >> sh) & 0xffff);
          else if (r.op == AIR_LBU) res = (c.m >> sh) & 0xff;
          else res = (uint32_t)(int32_t)(int8_t)((c.m >> sh) & 0xff);
          if (r.op != AIR_LW) out->sub_idx.push_back((uint32_t)(cycles - 1));
          ++rec.memory_ops;
          break;
        }
        case AIR_SB: case AIR_SH: case AIR_SW: {
          const uint32_t ad = a + r.imm;
          const int sz = (r.op == AIR_SW) ? 4 : (r.op == AIR_SH) ? 2 : 1;
          if (ad & (sz - 1)) FAULT(sz == 4 ? "unaligned sw" : "unaligned sh");
          CHECK_ADDR(ad, sz);
          if (ad < kRegSpace) FAULT("guest access below 0x10000 (register-mapped addresses)");
          const uint32_t w = ad & ~3u;
          c.m_pts = touch(w, ts + 2, 0, ((1u << sz) - 1) << (ad & 3));
          memcpy(&c.m, M + w, 4);
          const uint32_t sh = 8 * (ad & 3);
          if (r.op == AIR_SW) c.mv = b;
          else if (r.op == AIR_SH) c.mv = (c.m & ~(0xffffu << sh)) | ((b & 0xffff) << sh);
          else c.mv = (c.m & ~(0xffu << sh)) | ((b & 0xff) << sh);
          memcpy(M + w, &c.mv, 4);
          if (r.op != AIR_SW) out->sub_idx.push_back((uint32_t)(cycles - 1));
          ++rec.memory_ops;
          break;
        }
        case AIR_ECALL: {
          // t0 = a (code), a0 = b; a1 is read through the memory slot (register address 11)
          const uint32_t codeid = a, a0 = b, a1 = x[11], a2 = x[12];
          c.m = a1; c.mv = a1;
          c.m_pts = reg_ts[11]; reg_ts[11] = ts + 2;
          rec.syscall_counts[codeid & 0xff]++;
          out->ecall_idx.push_back((uint32_t)(cycles - 1));
          res = a;  // t0 is rewritten with itself except by HINT_LEN
          switch (codeid) {
            case 0x00: rec.exit_code = a0; rec.halted = true; halt = true; break;
            case 0x02: {
              CHECK_ADDR(a1, a2);
              const char* p = (const char*)(M + a1);
              if (a0 == 1) rec.stdout_text.append(p, a2);
              else if (a0 == 2) rec.stderr_text.append(p, a2);
              else if (a0 == 3) rec.public_values.insert(rec.public_values.end(), M + a1, M + a1 + a2);
              else if (a0 == 4) {}
              else FAULT("WRITE to unsupported fd");
              break;
            }
            case 0x10: if (a0 >= 8) FAULT("COMMIT word index out of range"); rec.pv_digest[a0] = a1; break;
            case 0x1a: if (a0 >= 8) FAULT("COMMIT_DEFERRED word index out of range"); rec.deferred_digest[a0] = a1; break;
            case 0xf0:
              if (stdin_pos >= stdin_entries.size()) FAULT("HINT_LEN: input stream exhausted");
              res = (uint32_t)stdin_entries[stdin_pos].size();
              break;
            case 0xf1: {
              if (stdin_pos >= stdin_entries.size()) FAULT("HINT_READ: input stream exhausted");
              const auto& e = stdin_entries[stdin_pos];
              if (a1 != e.size()) FAULT("HINT_READ: length mismatch");
              if (a0 & 3) FAULT("HINT_READ: unaligned pointer");
              if (a1 == 0 || a1 > (1u << 18) - 4) FAULT("HINT_READ: a read of no bytes, or of more than 2^18, has no rows in the hint chip");
              CHECK_ADDR(a0, (a1 + 3) & ~3u);
              // hinted words become the INITIAL memory contents of the proof (SP1 treats hint_read the same
              // way): they must not have been accessed before, and must lie outside the program image
              for (uint32_t w = a0; w < a0 + ((a1 + 3) & ~3u); w += 4)
                if (SH[w >> 2] != 0 || image_row(w) >= 0 || w < kRegSpace) FAULT("HINT_READ into memory that is already in use");
              memcpy(M + a0, e.data(), e.size());
              hinted.emplace_back(a0, a0 + ((a1 + 3) & ~3u));
              out->hint_words += (a1 + 3) / 4;
              ++stdin_pos;
              break;
            }
            default: FAULT("unsupported syscall code");
          }
          break;
        }
        case AIR_KECCAK: {
          const uint32_t ptr = b;  // a0
          if (ptr & 7) FAULT("keccakf state pointer not 8-byte aligned");
          CHECK_ADDR(ptr, 200);
          if (ptr < kRegSpace) FAULT("guest access below 0x10000 (register-mapped addresses)");
          KeccakCall k;
          k.ts = ts; k.ptr = ptr;
          memcpy(k.in, M + ptr, 200);
          for (int i = 0; i < 50; ++i) k.pts[i] = touch(ptr + 4 * (uint32_t)i, ts + 2, 0xf, 0xf);
          uint64_t st[25];
          memcpy(st, k.in, 200);
          keccak_f1600(st);
          memcpy(M + ptr, st, 200);
          out->keccak.push_back(k);
          KeccakEvent ev;
          memcpy(ev.state_in, k.in, 200);
          ev.state_ptr = ptr; ev.cycle = cycles - 1;
          rec.keccak_events.push_back(ev);
          next = a;  // return to ra
          break;
        }
        default: FAULT("internal: unknown AIR op");
      }
      c.a = res;
      if (alu_event) out->alu_idx.push_back((uint32_t)(cycles - 1));
      if (bw_event) out->bw_idx.push_back((uint32_t)(cycles - 1));
      if (r.wr) {
        c.w_prev = x[r.rd];
        c.w_pts = reg_ts[r.rd]; reg_ts[r.rd] = ts + 2;  // the written location is the row's third access
        x[r.rd] = res;
      }
      out->cycles.push_back(c);
      if (halt) break;
      pc = next;
    }
  }
done:
#undef FAULT
#undef CHECK_ADDR
  rec.cycles = cycles;
  if (!rec.error.empty()) return;
  // The CPU rows after the last cycle execute the padding instruction, which reads x0 once per row.  How many such rows
  // there are depends on the chip heights the run is proven with (a batch shares one shape), so the records stay
  // height-independent: the Program row of the padding instruction carries multiplicity 0 here and x0 is closed at
  // its last real access; whoever expands the records for given heights adds the padding rows' fetches and moves
  // x0's final time to the last row's (the trace expansion kernels).
  out->x0_last = reg_ts[0];
  // every image address (registers first: addresses 0..31) and every other touched address once, strictly increasing:
  // an untouched image word is closed with the value and the time (0) it was opened with
  std::sort(touched.begin(), touched.end(), [](const Touched& p, const Touched& q) { return p.addr < q.addr; });
  out->memfinal.reserve(prog.image.size() + touched.size());
  for (uint32_t i = 0; i < 32; ++i) out->memfinal.push_back({i, 0, x[i], reg_ts[i], 0});
  {
    size_t ti = 0;
    auto emit_touched = [&](const Touched& t) {
      uint32_t fin;
      memcpy(&fin, M + t.addr, 4);
      out->memfinal.push_back({t.addr, t.init, fin, t.last_ts, t.is_init});
    };
    for (size_t r = 32; r < prog.image.size(); ++r) {
      const ImageRow& im = prog.image[r];
      while (ti < touched.size() && touched[ti].addr < im.addr) emit_touched(touched[ti++]);
      if (ti < touched.size() && touched[ti].addr == im.addr) emit_touched(touched[ti++]);
      else out->memfinal.push_back({im.addr, im.val, im.val, 0, 0});
    }
    while (ti < touched.size()) emit_touched(touched[ti++]);
  }
}

void LeafCheckLog::append_all(const LeafCheckLog* const* parts, size_t n) {
  std::vector<size_t> at_p2(n + 1), at_qr(n + 1);
  at_p2[0] = p2_rows.size();
  at_qr[0] = qr_rows.size();
  for (size_t k = 0; k < n; ++k) {
    at_p2[k + 1] = at_p2[k] + parts[k]->p2_rows.size();
    at_qr[k + 1] = at_qr[k] + parts[k]->qr_rows.size();
    tr_rows.insert(tr_rows.end(), parts[k]->tr_rows.begin(), parts[k]->tr_rows.end());
    pub_tuples.insert(pub_tuples.end(), parts[k]->pub_tuples.begin(), parts[k]->pub_tuples.end());
  }
  p2_rows.resize(at_p2[n]);
  qr_rows.resize(at_qr[n]);
  auto copy = [&](size_t k) noexcept {
    if (!parts[k]->p2_rows.empty()) memcpy(p2_rows.data() + at_p2[k], parts[k]->p2_rows.data(), parts[k]->p2_rows.size() * 4);
    if (!parts[k]->qr_rows.empty()) memcpy(qr_rows.data() + at_qr[k], parts[k]->qr_rows.data(), parts[k]->qr_rows.size() * 4);
  };
  struct Joiner {
    std::vector<std::thread> th;
    ~Joiner() { for (auto& t : th) if (t.joinable()) t.join(); }
  } pool;
  for (size_t k = 1; k < n; ++k) {
    try {
      pool.th.emplace_back(copy, k);
    } catch (...) {
      copy(k);
    }
  }
  if (n) copy(0);
}

}  // namespace zksp
