// See executor.hpp.  Golden behaviour (cycle counts, public values, exit codes)
// is pinned by SURVEY.md appendix A.4 and checked in tests/test_executor.py.
#include "executor.hpp"

#include <sys/mman.h>

#include <cstdio>
#include <cstring>

namespace zksp {

// ---------------------------------------------------------------------------
// hashes
// ---------------------------------------------------------------------------
static const uint64_t kKeccakRC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
    0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
    0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
// rotation offset of lane (x, y), indexed [x][y]
static const int kKeccakRot[5][5] = {{0, 36, 3, 41, 18},
                                     {1, 44, 10, 45, 2},
                                     {62, 6, 43, 15, 61},
                                     {28, 55, 25, 21, 56},
                                     {27, 20, 39, 8, 14}};

static inline uint64_t rol64(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }

void keccak_f1600(uint64_t a[25]) {
  for (int rnd = 0; rnd < 24; ++rnd) {
    uint64_t c[5], d[5], b[25];
    for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
    for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
    for (int x = 0; x < 5; ++x)
      for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y], kKeccakRot[x][y]);
    for (int y = 0; y < 5; ++y)
      for (int x = 0; x < 5; ++x) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= kKeccakRC[rnd];
  }
}

void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  const size_t rate = 136;
  uint64_t st[25] = {0};
  uint8_t blk[136];
  size_t off = 0;
  for (;;) {
    size_t n = len - off < rate ? len - off : rate;
    bool last = n < rate;
    memset(blk, 0, rate);
    memcpy(blk, data + off, n);
    if (last) {
      blk[n] ^= 0x01;
      blk[rate - 1] ^= 0x80;
    }
    for (size_t i = 0; i < rate / 8; ++i) {
      uint64_t w;
      memcpy(&w, blk + 8 * i, 8);
      st[i] ^= w;
    }
    keccak_f1600(st);
    off += n;
    if (last) break;
  }
  memcpy(out, st, 32);
}

static inline uint32_t ror32(uint32_t v, int n) { return (v >> n) | (v << (32 - n)); }

void sha256(const uint8_t* data, size_t len, uint8_t out[32]) {
  static const uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
      0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
      0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
      0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
      0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
      0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
      0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  std::vector<uint8_t> msg(data, data + len);
  msg.push_back(0x80);
  while (msg.size() % 64 != 56) msg.push_back(0);
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 7; i >= 0; --i) msg.push_back((uint8_t)(bits >> (8 * i)));
  for (size_t off = 0; off < msg.size(); off += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i)
      w[i] = (uint32_t)msg[off + 4 * i] << 24 | (uint32_t)msg[off + 4 * i + 1] << 16 |
             (uint32_t)msg[off + 4 * i + 2] << 8 | msg[off + 4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      uint32_t s0 = ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3);
      uint32_t s1 = ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      uint32_t S1 = ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25);
      uint32_t ch = (e & f) ^ (~e & g);
      uint32_t t1 = hh + S1 + ch + K[i] + w[i];
      uint32_t S0 = ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22);
      uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
      uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  for (int i = 0; i < 8; ++i) {
    out[4 * i] = (uint8_t)(h[i] >> 24);
    out[4 * i + 1] = (uint8_t)(h[i] >> 16);
    out[4 * i + 2] = (uint8_t)(h[i] >> 8);
    out[4 * i + 3] = (uint8_t)h[i];
  }
}

// ---------------------------------------------------------------------------
// ELF loader
// ---------------------------------------------------------------------------
static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | p[1] << 8); }
static inline uint32_t rd32(const uint8_t* p) {
  return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}

void predecode(ElfImage* elf);

std::string load_elf(const uint8_t* d, size_t len, ElfImage* out) {
  if (len < 52 || memcmp(d, "\x7f" "ELF", 4) != 0) return "not an ELF file";
  if (d[4] != 1 || d[5] != 1) return "not ELF32 little-endian";
  if (rd16(d + 18) != 0xF3) return "not a RISC-V ELF";
  *out = ElfImage();
  out->entry = rd32(d + 24);
  uint32_t phoff = rd32(d + 28), shoff = rd32(d + 32);
  uint16_t phentsize = rd16(d + 42), phnum = rd16(d + 44);
  uint16_t shentsize = rd16(d + 46), shnum = rd16(d + 48);
  bool have_text = false;
  for (uint16_t i = 0; i < phnum; ++i) {
    size_t o = (size_t)phoff + (size_t)i * phentsize;
    if (o + 32 > len) return "program header out of range";
    const uint8_t* p = d + o;
    if (rd32(p) != 1) continue;  // PT_LOAD
    uint32_t off = rd32(p + 4), vaddr = rd32(p + 8), filesz = rd32(p + 16), memsz = rd32(p + 20), flags = rd32(p + 24);
    if ((size_t)off + filesz > len) return "segment out of range";
    if (vaddr & 3) return "unaligned segment";
    ElfImage::Seg s;
    s.vaddr = vaddr;
    s.memsz = memsz;
    s.bytes.assign(d + off, d + off + filesz);
    if (vaddr + memsz > out->max_addr) out->max_addr = vaddr + memsz;
    if ((flags & 1) && !have_text) {  // PF_X
      have_text = true;
      out->text_base = vaddr;
      out->text.resize((filesz + 3) / 4);
      for (size_t w = 0; w < out->text.size(); ++w) {
        uint8_t tmp[4] = {0, 0, 0, 0};
        size_t n = filesz - 4 * w < 4 ? filesz - 4 * w : 4;
        memcpy(tmp, d + off + 4 * w, n);
        out->text[w] = rd32(tmp);
      }
    }
    out->segs.push_back(std::move(s));
  }
  if (!have_text) return "no executable segment";
  // symbol table: FUNC symbols whose name contains "keccakf"
  for (uint16_t i = 0; i < shnum; ++i) {
    size_t o = (size_t)shoff + (size_t)i * shentsize;
    if (shoff == 0 || o + 40 > len) break;
    const uint8_t* sh = d + o;
    if (rd32(sh + 4) != 2) continue;  // SHT_SYMTAB
    uint32_t symoff = rd32(sh + 16), symsize = rd32(sh + 20), link = rd32(sh + 24), entsize = rd32(sh + 36);
    if (entsize < 16 || link >= shnum) continue;
    if ((size_t)shoff + (size_t)link * shentsize + 40 > len) continue;  // linked string-table header outside the file
    const uint8_t* strsh = d + shoff + (size_t)link * shentsize;
    uint32_t stroff = rd32(strsh + 16), strsize = rd32(strsh + 20);
    if ((size_t)symoff + symsize > len || (size_t)stroff + strsize > len) continue;
    for (uint32_t s = 0; s + entsize <= symsize; s += entsize) {
      const uint8_t* sym = d + symoff + s;
      uint32_t name = rd32(sym), value = rd32(sym + 4);
      uint8_t info = sym[12];
      if ((info & 0xf) != 2 || name >= strsize) continue;  // STT_FUNC
      const char* nm = (const char*)(d + stroff + name);
      size_t maxn = strsize - name;
      size_t nl = strnlen(nm, maxn);
      std::string sname(nm, nl);
      if (sname.find("keccakf") != std::string::npos && sname.find("closure") == std::string::npos) {
        bool dup = false;
        for (uint32_t e : out->keccakf_entries) dup |= (e == value);
        if (!dup) out->keccakf_entries.push_back(value);
      }
    }
  }
  uint8_t dg[32];
  sha256(d, len, dg);
  memcpy(out->sha256.data(), dg, 32);
  predecode(out);
  return "";
}

// ---------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------
struct Decoded {
  uint8_t op, rd, rs1, rs2;
  int32_t imm;
};

static Decoded decode(uint32_t w) {
  Decoded r{OP_INVALID, (uint8_t)((w >> 7) & 31), (uint8_t)((w >> 15) & 31), (uint8_t)((w >> 20) & 31), 0};
  uint32_t opc = w & 0x7f, f3 = (w >> 12) & 7, f7 = w >> 25;
  int32_t imm_i = (int32_t)w >> 20;
  int32_t imm_s = ((int32_t)(w & 0xfe000000) >> 20) | ((w >> 7) & 0x1f);
  int32_t imm_b = ((int32_t)(w & 0x80000000) >> 19) | ((w & 0x80) << 4) | ((w >> 20) & 0x7e0) | ((w >> 7) & 0x1e);
  int32_t imm_u = (int32_t)(w & 0xfffff000);
  int32_t imm_j = ((int32_t)(w & 0x80000000) >> 11) | (w & 0xff000) | ((w >> 9) & 0x800) | ((w >> 20) & 0x7fe);
  switch (opc) {
    case 0x37: r.op = OP_LUI; r.imm = imm_u; break;
    case 0x17: r.op = OP_AUIPC; r.imm = imm_u; break;
    case 0x6f: r.op = OP_JAL; r.imm = imm_j; break;
    case 0x67: if (f3 == 0) { r.op = OP_JALR; r.imm = imm_i; } break;
    case 0x63: {
      static const uint8_t m[8] = {OP_BEQ, OP_BNE, 0, 0, OP_BLT, OP_BGE, OP_BLTU, OP_BGEU};
      r.op = m[f3]; r.imm = imm_b; break;
    }
    case 0x03: {
      static const uint8_t m[8] = {OP_LB, OP_LH, OP_LW, 0, OP_LBU, OP_LHU, 0, 0};
      r.op = m[f3]; r.imm = imm_i; break;
    }
    case 0x23: {
      static const uint8_t m[8] = {OP_SB, OP_SH, OP_SW, 0, 0, 0, 0, 0};
      r.op = m[f3]; r.imm = imm_s; break;
    }
    case 0x13:
      r.imm = imm_i;
      switch (f3) {
        case 0: r.op = OP_ADDI; break;
        case 2: r.op = OP_SLTI; break;
        case 3: r.op = OP_SLTIU; break;
        case 4: r.op = OP_XORI; break;
        case 6: r.op = OP_ORI; break;
        case 7: r.op = OP_ANDI; break;
        case 1: if (f7 == 0) { r.op = OP_SLLI; r.imm = r.rs2; } break;
        case 5:
          if (f7 == 0) { r.op = OP_SRLI; r.imm = r.rs2; }
          else if (f7 == 0x20) { r.op = OP_SRAI; r.imm = r.rs2; }
          break;
      }
      break;
    case 0x33:
      if (f7 == 0) {
        static const uint8_t m[8] = {OP_ADD, OP_SLL, OP_SLT, OP_SLTU, OP_XOR, OP_SRL, OP_OR, OP_AND};
        r.op = m[f3];
      } else if (f7 == 0x20) {
        if (f3 == 0) r.op = OP_SUB;
        else if (f3 == 5) r.op = OP_SRA;
      } else if (f7 == 1) {
        static const uint8_t m[8] = {OP_MUL, OP_MULH, OP_MULHSU, OP_MULHU, OP_DIV, OP_DIVU, OP_REM, OP_REMU};
        r.op = m[f3];
      }
      break;
    case 0x0f: r.op = OP_FENCE; break;
    case 0x73:
      if (w == 0x00000073) r.op = OP_ECALL;
      else if (w == 0xc0001073) r.op = OP_UNIMP;
      break;
  }
  return r;
}

const char* op_name(int op) {
  static const char* n[OP_COUNT] = {
      "invalid", "lui", "auipc", "jal", "jalr", "beq", "bne", "blt", "bge", "bltu", "bgeu",
      "lb", "lh", "lw", "lbu", "lhu", "sb", "sh", "sw",
      "addi", "slti", "sltiu", "xori", "ori", "andi", "slli", "srli", "srai",
      "add", "sub", "sll", "slt", "sltu", "xor", "srl", "sra", "or", "and",
      "mul", "mulh", "mulhsu", "mulhu", "div", "divu", "rem", "remu",
      "ecall", "fence", "unimp"};
  return (op >= 0 && op < OP_COUNT) ? n[op] : "?";
}

// ---------------------------------------------------------------------------
// run loop
// ---------------------------------------------------------------------------
namespace {
// Guest address space: SP1 caps guest memory at 0x78000000; map it lazily.
constexpr uint64_t kMemBytes = 0x78000000ull;

struct Memory {
  uint8_t* base = nullptr;
  Memory() {
    void* p = mmap(nullptr, kMemBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    base = (p == MAP_FAILED) ? nullptr : (uint8_t*)p;
  }
  ~Memory() {
    if (base) munmap(base, kMemBytes);
  }
};
}  // namespace

// internal op id of the sentinel behind the last instruction
constexpr uint8_t kOpEndOfText = OP_COUNT;
constexpr uint8_t kHookBit = 0x80;

void predecode(ElfImage* elf) {
  elf->code.resize(elf->text.size() + 1);
  for (size_t i = 0; i < elf->text.size(); ++i) {
    const Decoded d = decode(elf->text[i]);
    elf->code[i] = {d.op, (uint8_t)(d.rd ? d.rd : 32), d.rs1, d.rs2, d.imm};
  }
  // the sentinel takes the rare path too: its fault precedes the cycle-limit check, as a fetch fault does
  elf->code[elf->text.size()] = {(uint8_t)(kOpEndOfText | kHookBit), 32, 0, 0, 0};
  const uint32_t lo = elf->text_base, hi = elf->text_base + 4 * (uint32_t)elf->text.size();
  for (uint32_t e : elf->keccakf_entries)
    if (e >= lo && e < hi && !(e & 3)) elf->code[(e - lo) >> 2].op |= kHookBit;
}

namespace {

// The run loop.  Threaded dispatch (one indirect jump per handler), the program counter kept as
// an index into the decoded text and range-checked only where control is transferred, x0 writes
// sent to a sink register, the keccakf hook carried by the decoded op.  Cycle, memory-operation
// and histogram accounting and every fault message are those of the plain switch interpreter
// this replaces (tests/test_executor.py pins them against SURVEY.md appendix A.4).
template <bool HIST>
void run(const ElfImage& elf, const std::vector<std::vector<uint8_t>>& stdin_entries, const ExecOptions& opt,
         uint8_t* M, ExecutionRecord& rec) {
  const ElfImage::Insn* code = elf.code.data();
  const uint32_t text_lo = elf.text_base, text_bytes = 4 * (uint32_t)elf.text.size();
  const bool hooks = opt.keccak_mode != KeccakMode::kSoftware;
  const bool replace = opt.keccak_mode == KeccakMode::kReplace;
  uint32_t x[33] = {0};
  size_t stdin_pos = 0;
  uint64_t cycles = 0, memops = 0;
  const uint64_t max_cycles = opt.max_cycles;
  size_t idx = 0;
  const ElfImage::Insn* in = code;
  uint32_t a = 0, b = 0, pc = 0;

  static const void* const kTable[OP_COUNT + 1] = {
      &&op_invalid, &&op_lui, &&op_auipc, &&op_jal, &&op_jalr, &&op_beq, &&op_bne, &&op_blt, &&op_bge, &&op_bltu,
      &&op_bgeu, &&op_lb, &&op_lh, &&op_lw, &&op_lbu, &&op_lhu, &&op_sb, &&op_sh, &&op_sw, &&op_addi, &&op_slti,
      &&op_sltiu, &&op_xori, &&op_ori, &&op_andi, &&op_slli, &&op_srli, &&op_srai, &&op_add, &&op_sub, &&op_sll,
      &&op_slt, &&op_sltu, &&op_xor, &&op_srl, &&op_sra, &&op_or, &&op_and, &&op_mul, &&op_mulh, &&op_mulhsu,
      &&op_mulhu, &&op_div, &&op_divu, &&op_rem, &&op_remu, &&op_ecall, &&op_fence, &&op_unimp, &&op_end_of_text};

#define FAULT(msg)     \
  do {                 \
    rec.error = (msg); \
    goto done;         \
  } while (0)
#define CHECK_ADDR(ad, n) \
  if ((uint64_t)(ad) + (n) > kMemBytes) FAULT("memory access out of range")
// transfer to guest address t (taken branch, jump): the fetch check of the next instruction
#define JUMP(t)                                                                      \
  do {                                                                               \
    const uint32_t t_ = (uint32_t)(t), off_ = t_ - text_lo;                          \
    if (off_ >= text_bytes || (t_ & 3)) { rec.fault_target = t_; FAULT("pc outside text segment"); } \
    idx = off_ >> 2;                                                                 \
  } while (0)
// fetch + dispatch, replicated in every handler so each has its own indirect jump
#define FETCH()                                                                      \
  do {                                                                               \
    in = &code[idx];                                                                 \
    op = in->op;                                                                     \
    if (__builtin_expect(op & kHookBit, 0)) goto rare;                               \
    if (__builtin_expect(cycles >= max_cycles, 0)) FAULT("cycle limit exceeded");    \
    ++cycles;                                                                        \
    if (HIST) rec.opcode_hist[op]++;                                                 \
    a = x[in->rs1];                                                                  \
    b = x[in->rs2];                                                                  \
    goto* kTable[op];                                                                \
  } while (0)
#define NEXT() \
  do {         \
    ++idx;     \
    FETCH();   \
  } while (0)

  uint8_t op = 0;
  if (opt.call_pc) {  // one function of the guest: it returns to the address behind the text
    x[10] = opt.call_a0;
    x[2] = opt.call_sp;
    x[1] = text_lo + text_bytes;
    JUMP(opt.call_pc);
  } else {
    JUMP(elf.entry);
  }
  FETCH();

rare:  // a keccakf entry point, or the sentinel behind the last instruction
  op &= (uint8_t)~kHookBit;
  if (op == kOpEndOfText) FAULT("pc outside text segment");
  if (hooks) {
    const uint32_t ptr = x[10];
    if (ptr & 7) FAULT("keccakf state pointer not 8-byte aligned");
    CHECK_ADDR(ptr, 200);
    KeccakEvent ev;
    memcpy(ev.state_in, M + ptr, 200);
    ev.state_ptr = ptr;
    ev.cycle = cycles;
    rec.keccak_events.push_back(ev);
    if (replace) {
      uint64_t st[25];
      memcpy(st, ev.state_in, 200);
      keccak_f1600(st);
      memcpy(M + ptr, st, 200);
      ++cycles;  // the precompile ecall; control returns to the caller
      JUMP(x[1]);
      FETCH();
    }
  }
  if (__builtin_expect(cycles >= max_cycles, 0)) FAULT("cycle limit exceeded");
  ++cycles;
  if (HIST) rec.opcode_hist[op]++;
  a = x[in->rs1];
  b = x[in->rs2];
  goto* kTable[op];

op_lui: x[in->rd] = (uint32_t)in->imm; NEXT();
op_auipc: x[in->rd] = text_lo + 4 * (uint32_t)idx + (uint32_t)in->imm; NEXT();
op_jal:
  pc = text_lo + 4 * (uint32_t)idx;
  x[in->rd] = pc + 4;
  JUMP(pc + (uint32_t)in->imm);
  FETCH();
op_jalr:
  pc = text_lo + 4 * (uint32_t)idx;
  x[in->rd] = pc + 4;
  JUMP((a + (uint32_t)in->imm) & ~1u);
  FETCH();
#define BRANCH(cond)                                              \
  if (cond) {                                                     \
    JUMP(text_lo + 4 * (uint32_t)idx + (uint32_t)in->imm);        \
    FETCH();                                                      \
  }                                                               \
  NEXT()
op_beq: BRANCH(a == b);
op_bne: BRANCH(a != b);
op_blt: BRANCH((int32_t)a < (int32_t)b);
op_bge: BRANCH((int32_t)a >= (int32_t)b);
op_bltu: BRANCH(a < b);
op_bgeu: BRANCH(a >= b);
#undef BRANCH
op_lb: { const uint32_t ad = a + (uint32_t)in->imm; CHECK_ADDR(ad, 1); x[in->rd] = (uint32_t)(int32_t)(int8_t)M[ad]; ++memops; NEXT(); }
op_lbu: { const uint32_t ad = a + (uint32_t)in->imm; CHECK_ADDR(ad, 1); x[in->rd] = M[ad]; ++memops; NEXT(); }
op_lh: { const uint32_t ad = a + (uint32_t)in->imm; if (ad & 1) FAULT("unaligned lh"); CHECK_ADDR(ad, 2); uint16_t v; memcpy(&v, M + ad, 2); x[in->rd] = (uint32_t)(int32_t)(int16_t)v; ++memops; NEXT(); }
op_lhu: { const uint32_t ad = a + (uint32_t)in->imm; if (ad & 1) FAULT("unaligned lhu"); CHECK_ADDR(ad, 2); uint16_t v; memcpy(&v, M + ad, 2); x[in->rd] = v; ++memops; NEXT(); }
op_lw: { const uint32_t ad = a + (uint32_t)in->imm; if (ad & 3) FAULT("unaligned lw"); CHECK_ADDR(ad, 4); uint32_t v; memcpy(&v, M + ad, 4); x[in->rd] = v; ++memops; NEXT(); }
op_sb: { const uint32_t ad = a + (uint32_t)in->imm; CHECK_ADDR(ad, 1); M[ad] = (uint8_t)b; ++memops; NEXT(); }
op_sh: { const uint32_t ad = a + (uint32_t)in->imm; if (ad & 1) FAULT("unaligned sh"); CHECK_ADDR(ad, 2); const uint16_t v = (uint16_t)b; memcpy(M + ad, &v, 2); ++memops; NEXT(); }
op_sw: { const uint32_t ad = a + (uint32_t)in->imm; if (ad & 3) FAULT("unaligned sw"); CHECK_ADDR(ad, 4); memcpy(M + ad, &b, 4); ++memops; NEXT(); }
op_addi: x[in->rd] = a + (uint32_t)in->imm; NEXT();
op_slti: x[in->rd] = (int32_t)a < in->imm; NEXT();
op_sltiu: x[in->rd] = a < (uint32_t)in->imm; NEXT();
op_xori: x[in->rd] = a ^ (uint32_t)in->imm; NEXT();
op_ori: x[in->rd] = a | (uint32_t)in->imm; NEXT();
op_andi: x[in->rd] = a & (uint32_t)in->imm; NEXT();
op_slli: x[in->rd] = a << in->imm; NEXT();
op_srli: x[in->rd] = a >> in->imm; NEXT();
op_srai: x[in->rd] = (uint32_t)((int32_t)a >> in->imm); NEXT();
op_add: x[in->rd] = a + b; NEXT();
op_sub: x[in->rd] = a - b; NEXT();
op_sll: x[in->rd] = a << (b & 31); NEXT();
op_slt: x[in->rd] = (int32_t)a < (int32_t)b; NEXT();
op_sltu: x[in->rd] = a < b; NEXT();
op_xor: x[in->rd] = a ^ b; NEXT();
op_srl: x[in->rd] = a >> (b & 31); NEXT();
op_sra: x[in->rd] = (uint32_t)((int32_t)a >> (b & 31)); NEXT();
op_or: x[in->rd] = a | b; NEXT();
op_and: x[in->rd] = a & b; NEXT();
op_mul: x[in->rd] = a * b; NEXT();
op_mulh: x[in->rd] = (uint32_t)(((int64_t)(int32_t)a * (int64_t)(int32_t)b) >> 32); NEXT();
op_mulhsu: x[in->rd] = (uint32_t)(((int64_t)(int32_t)a * (int64_t)(uint64_t)b) >> 32); NEXT();
op_mulhu: x[in->rd] = (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); NEXT();
op_div:
  if (b == 0) x[in->rd] = 0xffffffffu;
  else if (a == 0x80000000u && b == 0xffffffffu) x[in->rd] = a;
  else x[in->rd] = (uint32_t)((int32_t)a / (int32_t)b);
  NEXT();
op_divu: x[in->rd] = b ? a / b : 0xffffffffu; NEXT();
op_rem:
  if (b == 0) x[in->rd] = a;
  else if (a == 0x80000000u && b == 0xffffffffu) x[in->rd] = 0;
  else x[in->rd] = (uint32_t)((int32_t)a % (int32_t)b);
  NEXT();
op_remu: x[in->rd] = b ? a % b : a; NEXT();
op_fence: NEXT();
op_ecall: {
  const uint32_t codeid = x[5], a0 = x[10], a1 = x[11], a2 = x[12];
  rec.syscall_counts[codeid & 0xff]++;
  switch (codeid) {
    case 0x00:  // HALT
      rec.exit_code = a0;
      rec.halted = true;
      goto done;
    case 0x02: {  // WRITE
      CHECK_ADDR(a1, a2);
      const char* p = (const char*)(M + a1);
      if (a0 == 1) rec.stdout_text.append(p, a2);
      else if (a0 == 2) rec.stderr_text.append(p, a2);
      else if (a0 == 3) rec.public_values.insert(rec.public_values.end(), M + a1, M + a1 + a2);
      else if (a0 == 4) { /* hint-stream write: not used by this guest */ }
      else FAULT("WRITE to unsupported fd");
      break;
    }
    case 0x10:  // COMMIT
      if (a0 >= 8) FAULT("COMMIT word index out of range");
      rec.pv_digest[a0] = a1;
      break;
    case 0x1a:  // COMMIT_DEFERRED_PROOFS
      if (a0 >= 8) FAULT("COMMIT_DEFERRED word index out of range");
      rec.deferred_digest[a0] = a1;
      break;
    case 0xf0:  // HINT_LEN
      if (stdin_pos >= stdin_entries.size()) FAULT("HINT_LEN: input stream exhausted");
      x[5] = (uint32_t)stdin_entries[stdin_pos].size();
      break;
    case 0xf1: {  // HINT_READ
      if (stdin_pos >= stdin_entries.size()) FAULT("HINT_READ: input stream exhausted");
      const auto& e = stdin_entries[stdin_pos];
      if (a1 != e.size()) FAULT("HINT_READ: length mismatch");
      if (a0 & 3) FAULT("HINT_READ: unaligned pointer");
      CHECK_ADDR(a0, (a1 + 3) & ~3u);
      memcpy(M + a0, e.data(), e.size());
      ++stdin_pos;
      break;
    }
    default:
      FAULT("unsupported syscall code");
  }
  NEXT();
}
op_unimp: FAULT("unimp executed");
op_invalid: FAULT("illegal instruction");
op_end_of_text:  // not reachable through the table (the sentinel takes the rare path)
  FAULT("pc outside text segment");

done:
#undef FAULT
#undef CHECK_ADDR
#undef JUMP
#undef FETCH
#undef NEXT
  rec.cycles = cycles;
  rec.memory_ops = memops;
}

}  // namespace

ExecutionRecord execute(const ElfImage& elf, const std::vector<std::vector<uint8_t>>& stdin_entries,
                        const ExecOptions& opt) {
  ExecutionRecord rec;
  if (opt.want_hist) rec.opcode_hist.assign(OP_COUNT + 1, 0);  // + the sentinel's slot, dropped below
  Memory mem;
  if (!mem.base) {
    rec.error = "mmap of guest memory failed";
    return rec;
  }
  uint8_t* M = mem.base;
  for (const auto& s : elf.segs) {
    const uint64_t extent = s.bytes.size() > s.memsz ? s.bytes.size() : s.memsz;  // a hostile p_filesz > p_memsz
    if ((uint64_t)s.vaddr + extent > kMemBytes) {
      rec.error = "segment beyond guest memory";
      return rec;
    }
    if (!s.bytes.empty()) memcpy(M + s.vaddr, s.bytes.data(), s.bytes.size());
  }
  if (elf.code.size() != elf.text.size() + 1) {
    rec.error = "ELF image was not decoded (load_elf)";
    return rec;
  }
  if (opt.call_pc) {
    if ((opt.call_a0 & 7) || (uint64_t)opt.call_a0 + 200 > kMemBytes || !opt.call_state) {
      rec.error = "call: bad state pointer";
      return rec;
    }
    memcpy(M + opt.call_a0, opt.call_state, 200);
  }
  if (opt.want_hist) run<true>(elf, stdin_entries, opt, M, rec);
  else run<false>(elf, stdin_entries, opt, M, rec);
  if (opt.want_hist) rec.opcode_hist.resize(OP_COUNT);
  if (opt.call_pc) memcpy(rec.call_state_out, M + opt.call_a0, 200);
  return rec;
}

std::string check_keccakf_entries(const ElfImage& elf) {
  // the zero state, a state of all ones, and two states from a fixed xorshift stream
  uint64_t vec[4][25];
  uint64_t z = 0x9e3779b97f4a7c15ull;
  for (int i = 0; i < 25; ++i) {
    vec[0][i] = 0;
    vec[1][i] = ~0ull;
    for (int v = 2; v < 4; ++v) {
      z ^= z << 13; z ^= z >> 7; z ^= z << 17;
      vec[v][i] = z;
    }
  }
  const uint32_t text_end = elf.text_base + 4 * (uint32_t)elf.text.size();
  for (uint32_t e : elf.keccakf_entries) {
    char where[32];
    snprintf(where, sizeof where, "0x%x", e);
    for (int v = 0; v < 4; ++v) {
      ExecOptions o;
      o.keccak_mode = KeccakMode::kSoftware;
      o.max_cycles = 1u << 22;  // (a software keccak-f is some 30 000 cycles)
      o.call_pc = e;
      o.call_a0 = (uint32_t)(kMemBytes - 4096);
      o.call_sp = (uint32_t)(kMemBytes - 65536);
      o.call_state = vec[v];
      const ExecutionRecord r = execute(elf, {}, o);
      if (r.error != "pc outside text segment" || r.fault_target != text_end)
        return std::string("the function at ") + where + " (named keccakf) did not return from a test call: " + r.error;
      uint64_t want[25];
      memcpy(want, vec[v], 200);
      keccak_f1600(want);
      if (memcmp(want, r.call_state_out, 200) != 0)
        return std::string("the function at ") + where + " is named keccakf but does not compute keccak-f[1600] on the test states";
    }
  }
  return "";
}

}  // namespace zksp
