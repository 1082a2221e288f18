// Chips of the machine proof: column layouts and base-field
// constraints, written once as field-generic templates.  The device quotient kernels instantiate
// them over Fp (one lane = one LDE-domain point), the host verifier over Fp4 (the point zeta).
//
// This is this repository's own arithmetisation of RV32IM + the keccak precompile; it stands where
// sp1-core-machine 3.4.0's CPU / memory / program / ALU / keccak-permute chips stand beneath the
// reference's `client.prove(&pk, stdin).run()` (prover/src/bin/main.rs:71-74, Cargo.lock:7130).
// What it must establish is the reference's statement: the committed guest
// (circuits/sp1-merkle-proof/src/main.rs:4-14 running crypto-ops/src/lib.rs:8-23) executed from its
// entry point to HALT with the committed public values.  DESIGN.md "Machine proof" describes the
// construction; constraint ORDER here is normative for the proof bytes.
#pragma once
#include "air_keccak.hpp"

namespace zksp {
namespace mach {

// The execution is split over two instances of the CPU chip: cycles [0, H0) in kCpu, H0 the largest power of two below
// the cycle count, the rest in kCpu2 (a power of two again): 391 400 cycles take 2^18 + 2^17 rows instead of 2^19.
enum Chip { kCpu = 0, kKeccak, kKmem, kMemFinal, kImage, kProgram, kMul, kRange, kCpu2, kNumChips };
// public scalars of a CPU instance: pc and time of its first row, whether another instance continues it, and the pc
// that one starts at (the hand-over pc: a proof-header word the transcript absorbs)
enum CpuPub { kPubStartPc = 0, kPubStartTs, kPubHasSucc, kPubEndPc, kNumCpuPub };
ZKSP_HD constexpr bool is_cpu_chip(int chip) { return chip == kCpu || chip == kCpu2; }

// AIR opcodes = Program-table column OP = 1 + index of the CPU selector column
enum Op {
  ADD = 1, SUB, XOR, OR, AND, SLL, SRL, SRA, SLT, SLTU, JAL, JALR, BEQ, BNE, BLT, BGE, BLTU, BGEU, LB, LH, LW, LBU, LHU,
  SB, SH, SW, MUL, MULHU, ECALL, KECCAK
};
constexpr int kNumOps = 30;
// access-time differences: two limbs of kTsLimbBits bits, each looked up in the range table
constexpr int kTsLimbBits = 12, kTsLimbs = 2;

// ---- CPU chip ----
constexpr int C_IS_REAL = 0, C_PC = 1, C_TS = 2, C_NEXT_PC = 3, C_OP = 4, C_WR = C_OP + kNumOps, C_USE2 = C_WR + 1,
              C_RD = C_WR + 2, C_RS1 = C_WR + 3, C_RS2 = C_WR + 4, C_IMM_LO = C_WR + 5, C_IMM_HI = C_WR + 6, C_TGT = C_WR + 7,
              C_A = C_WR + 8, C_B = C_A + 2, C_C = C_B + 32, C_M = C_C + 32, C_X = C_M + 32, C_MV_LO = C_X + 32,
              C_MV_HI = C_MV_LO + 1, C_K0 = C_MV_LO + 2, C_K1 = C_K0 + 1, C_K2 = C_K0 + 2, C_K3 = C_K0 + 3, C_EQ = C_K0 + 4,
              C_INV = C_K0 + 5, C_O0 = C_K0 + 6, C_O1 = C_O0 + 1, C_O2 = C_O0 + 2, C_O3 = C_O0 + 3, C_SC = C_O0 + 4,
              C_R1_PTS = C_SC + 6, C_R2_PTS = C_R1_PTS + 1, C_M_PTS = C_R1_PTS + 2, C_W_PTS = C_R1_PTS + 3,
              C_W_PLO = C_R1_PTS + 4, C_W_PHI = C_R1_PTS + 5, C_R1_D = C_R1_PTS + 6, C_R2_D = C_R1_D + kTsLimbs,
              C_M_D = C_R2_D + kTsLimbs, C_W_D = C_M_D + kTsLimbs, kCpuWidth = C_W_D + kTsLimbs;
enum { SC_HALT = 0, SC_WRITE, SC_COMMIT, SC_DEFER, SC_HINT_LEN, SC_HINT_READ };
static_assert(kCpuWidth == 204, "CPU chip layout");

// ---- keccak chip: p3-keccak-air's columns + the call time ----
constexpr int KC_TS = ka::kWidth, kKeccakWidth = ka::kWidth + 1;
// ---- keccak-memory chip ----
constexpr int KM_IS_REAL = 0, KM_TS = 1, KM_PTR_LO = 2, KM_PTR_HI = 3, KM_IDX = 4, KM_ISF = 5, KM_ISL = 6, KM_CALL = 7,
              KM_ADDR = 8, KM_OLD_LO = 9, KM_OLD_HI = 10, KM_NEW_LO = 11, KM_NEW_HI = 12, KM_PTS = 13, KM_D = 14,
              kKmemWidth = KM_D + kTsLimbs;
// ---- memory boundary chip ----
constexpr int MF_IS_REAL = 0, MF_ADDR = 1, MF_IS_INIT = 2, MF_FIN_LO = 3, MF_FIN_HI = 4, MF_FIN_TS = 5, MF_DIFF = 6,
              MF_INIT = MF_DIFF + 32, kMemFinalWidth = MF_INIT + 32;
// ---- image / program chips: preprocessed columns, one main column ----
constexpr int IMG_P_ADDR = 0, IMG_P_LO = 1, IMG_P_HI = 2, kImagePrepWidth = 3, kImageWidth = 1;
constexpr int PR_PC = 0, PR_OP = 1, PR_WR = 2, PR_USE2 = 3, PR_RD = 4, PR_RS1 = 5, PR_RS2 = 6, PR_IMM_LO = 7, PR_IMM_HI = 8,
              PR_TGT = 9, kProgramPrepWidth = 10, kProgramWidth = 1;
// ---- multiplier chip ----
constexpr int MU_IS_REAL = 0, MU_HI = 1, MU_B = 2, MU_C = MU_B + 32, MU_P = MU_C + 32, MU_Q0 = MU_P + 64, MU_Q1 = MU_Q0 + 10,
              MU_Q2 = MU_Q1 + 11, kMulWidth = MU_Q2 + 10;

// ---- range table: preprocessed (value = row index), main (multiplicity); always 2^kTsLimbBits rows ----
constexpr int kRangePrepWidth = 1, kRangeWidth = 1, kRangeLogH = kTsLimbBits;

enum Bus { BUS_MEM = 1, BUS_PROG, BUS_KCALL, BUS_KIO, BUS_MUL, BUS_PUBC, BUS_PUBH, BUS_RANGE };

// Ctx interface:
//   using F;  F local(int col); F next(int col); F is_first(); F is_trans(); F is_last(); F pub(int which)  (CpuPub);
//   F k(uint32_t montgomery_word)  (a constant);  void emit(F v)  (appends the next constraint);
//   void emit_at(int index, F v);  void set_count(int n)  (index of the next emit());
//   F sum_prod(const F* x, const F* y, int ystep, int n)  (eval_cpu only)
//   void stash(int i, F v); F stashed(int i)  (eval_cpu only): 32 values parked by index and read back by index (the
//     shift constraints revisit B's bits in a loop the device compiler keeps rolled: the device parks them in LDS
//     instead of going back to HBM three more times per bit)
//   void note_limbs(int block, F lo, F hi)  (eval_cpu only): the 16-bit limbs of bit block B (0), C (1), M (2), X (3)
//     as the task that streams the block has them; the device keeps them so that the LogUp task need not read the bits
#define ZKSP_K(c) ctx.k(cmonty(c))

template <class F, class Ctx>
ZKSP_HD F limb_of(const Ctx& ctx, int bits, int limb) {
  F s = ctx.local(bits + 16 * limb + 15);
  for (int i = 14; i >= 0; --i) s = s.dbl() + ctx.local(bits + 16 * limb + i);
  return s;
}
template <class F, class Ctx>
ZKSP_HD F byte_of(const Ctx& ctx, int bits, int byte) {
  F s = ctx.local(bits + 8 * byte + 7);
  for (int i = 6; i >= 0; --i) s = s.dbl() + ctx.local(bits + 8 * byte + i);
  return s;
}
template <class F, class Ctx>
ZKSP_HD F bits_val(const Ctx& ctx, int bits, int n) {
  F s = ctx.local(bits + n - 1);
  for (int i = n - 2; i >= 0; --i) s = s.dbl() + ctx.local(bits + i);
  return s;
}
template <class F>
ZKSP_HD F bool_c(F v, F one) {
  return v * (v - one);
}

// Constraint index space of the CPU chip (fixes which power of alpha multiplies which constraint):
//   0..175 booleans (IS_REAL, OP[30], WR, USE2, B/C/M/X bits, K0..3, EQ, O0..3, SC[6]),
//   176..185 row structure, 186..187 immediate operand, 188..191 add/sub, 192..197 xor/or/and,
//   198..205 shifts, 206..211 comparisons, 212..224 next pc, 225..226 address adder, 227..232 byte
//   offset, 233..255 loads/stores, 256..259 ecall, 260..263 access times, 264..265 hand-over to the next instance.
// The evaluation below walks the columns block by block (each column is read once, its block's
// arrays die before the next block is loaded) and emits by index, so the device kernel keeps a few
// dozen live values instead of reloading 5 000 operands per point.
// Ctx additionally provides  F sum_prod(const F* x, const F* y, int ystep, int n) = sum x[i] * y[i * ystep].
namespace cpuidx {
constexpr int kBoolB = 33, kBoolC = 65, kBoolM = 97, kBoolX = 129, kBoolK = 161, kBoolEq = 165, kBoolO = 166,
              kBoolSc = 170, kStruct = 176, kImm = 186, kAddSub = 188, kBitwise = 192, kShift = 198, kCmp = 206,
              kNextPc = 212, kAddr = 225, kOff = 227, kLoadStore = 233, kEcall = 256, kTimes = 260, kHandOver = 264;
}

ZKSP_HD constexpr uint32_t pow2_mod(int n) { return (uint32_t)(((uint64_t)1 << n) % kP); }
ZKSP_HD constexpr uint32_t inv_pow2_mod(int n) {  // 2^-n mod p
  uint64_t r = 1;
  for (int i = 0; i < n; ++i) r = r * ((kP + 1) / 2) % kP;
  return (uint32_t)r;
}

template <class F>
ZKSP_HD F limb16(const F* bits, int limb) {
  F s = bits[16 * limb + 15];
#pragma unroll
  for (int i = 14; i >= 0; --i) s = s.dbl() + bits[16 * limb + i];
  return s;
}
template <class F>
ZKSP_HD F byte8(const F* bits, int byte) {
  F s = bits[8 * byte + 7];
#pragma unroll
  for (int i = 6; i >= 0; --i) s = s.dbl() + bits[8 * byte + i];
  return s;
}

// The 266 constraints in four independent tasks, each reading only the column blocks it needs (a block
// that two tasks need is read by both): the device runs a task per workgroup, so a lane holds a few
// dozen live values instead of the whole 204-column row; the verifier runs all four in sequence.
//   task 0  selectors, row structure, the four access-time differences           (scalars)
//   task 1  A, B, C: immediate operand, add / sub, bitwise, jal / jalr link, ecall, keccak return
//   task 2  X with A, B, C: shifts, comparisons, branches, jalr target, address adder, byte offset
//   task 3  M with A, C: loads and stores
constexpr int kCpuTasks = 4;

template <class F, class Ctx>
ZKSP_HD void load_bits(Ctx& ctx, int col, F* out) {
#pragma unroll
  for (int i = 0; i < 32; ++i) out[i] = ctx.local(col + i);
}
template <class F, class Ctx>
ZKSP_HD void limbs_of_block(Ctx& ctx, int col, F* lo, F* hi) {  // streams the block: no array kept
  F l = ctx.local(col + 15), h = ctx.local(col + 31);
#pragma unroll
  for (int i = 14; i >= 0; --i) {
    l = l.dbl() + ctx.local(col + i);
    h = h.dbl() + ctx.local(col + 16 + i);
  }
  *lo = l;
  *hi = h;
}

template <int TASK, class Ctx>
ZKSP_HD void eval_cpu_task(Ctx& ctx) {
  using F = typename Ctx::F;
  using namespace cpuidx;
  const F one = ctx.k(kR1), zero = one - one;
  const F k65536 = ZKSP_K(65536);
#define L(c) ctx.local(c)
#define OPF(o) ctx.local(C_OP + (o) - 1)
  if (TASK == 0) {
    const F is_real = L(C_IS_REAL);
    ctx.emit_at(0, bool_c(is_real, one));
    F opsum = zero, ld_st_ecall = zero;
#pragma unroll
    for (int k = 1; k <= kNumOps; ++k) {
      const F o = L(C_OP + k - 1);
      ctx.emit_at(k, bool_c(o, one));
      opsum = opsum + o;
      if ((k >= LB && k <= SW) || k == ECALL) ld_st_ecall = ld_st_ecall + o;
    }
    const F wr = L(C_WR), use2 = L(C_USE2), is_first = ctx.is_first(), is_trans = ctx.is_trans();
    ctx.emit_at(31, bool_c(wr, one));
    ctx.emit_at(32, bool_c(use2, one));
    const F pc = L(C_PC), ts = L(C_TS), np = L(C_NEXT_PC);
    F scsum = zero, sc_halt = zero;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const F v = L(C_SC + k);
      ctx.emit_at(kBoolSc + k, bool_c(v, one));
      scsum = scsum + v;
      if (k == SC_HALT) sc_halt = v;
    }
    ctx.emit_at(kStruct + 0, opsum - is_real);
    ctx.emit_at(kStruct + 1, wr * (one - is_real));
    ctx.emit_at(kStruct + 2, use2 * (one - is_real));
    ctx.emit_at(kStruct + 3, is_first * (is_real - one));
    ctx.emit_at(kStruct + 4, is_first * (pc - ctx.pub(kPubStartPc)));
    ctx.emit_at(kStruct + 5, is_first * (ts - ctx.pub(kPubStartTs)));
    ctx.emit_at(kStruct + 6, is_trans * (ctx.next(C_TS) - ts - ZKSP_K(4)));
    const F nreal = ctx.next(C_IS_REAL);
    ctx.emit_at(kStruct + 7, is_trans * nreal * (ctx.next(C_PC) - np));
    ctx.emit_at(kStruct + 8, is_trans * (nreal - is_real + sc_halt));
    ctx.emit_at(kStruct + 9, scsum - OPF(ECALL));
#pragma unroll
    for (int i = 0; i < 4; ++i) ctx.emit_at(kBoolK + i, bool_c(L(C_K0 + i), one));
    ctx.emit_at(kBoolEq, bool_c(L(C_EQ), one));
#pragma unroll
    for (int i = 0; i < 4; ++i) ctx.emit_at(kBoolO + i, bool_c(L(C_O0 + i), one));
    // access times: a difference is limb 0 + 2^12 limb 1; the limbs' ranges come from the RANGE bus
    const int dcol[4] = {C_R1_D, C_R2_D, C_M_D, C_W_D};
    const F klimb = ZKSP_K(1u << kTsLimbBits);
    F dv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) dv[q] = L(dcol[q]) + klimb * L(dcol[q] + 1);
    ctx.emit_at(kTimes + 0, is_real * (ts - L(C_R1_PTS) - one - dv[0]));
    ctx.emit_at(kTimes + 1, use2 * (ts - L(C_R2_PTS) - dv[1]));
    ctx.emit_at(kTimes + 2, ld_st_ecall * (ts + one - L(C_M_PTS) - dv[2]));
    ctx.emit_at(kTimes + 3, wr * (ts + ZKSP_K(2) - L(C_W_PTS) - dv[3]));
    // hand-over to the next instance: its last row is a real row that does not halt and names the pc the next
    // instance starts at
    const F succ = ctx.is_last() * ctx.pub(kPubHasSucc);
    ctx.emit_at(kHandOver + 0, succ * (one - is_real + sc_halt));
    ctx.emit_at(kHandOver + 1, succ * (L(C_NEXT_PC) - ctx.pub(kPubEndPc)));
  }
  if (TASK == 1) {
    // B and C bit by bit (top bit first): limbs by Horner, the three bitwise results per half.  A is its two limbs:
    // whatever is read back from a register or from memory is read through bits (B, C, M), so a written value
    // whose limbs were out of range could never be consumed; it needs no range check of its own.
    const F a_lo = L(C_A), a_hi = L(C_A + 1);
    F b_lo = zero, b_hi = zero, c_lo = zero, c_hi = zero;
    F ax[2] = {zero, zero}, ao[2] = {zero, zero}, aa[2] = {zero, zero};
#pragma unroll
    for (int h = 1; h >= 0; --h) {
      for (int i = 15; i >= 0; --i) {
        const int col = 16 * h + i;
        const F b = L(C_B + col), c = L(C_C + col);
        ctx.emit_at(kBoolB + col, bool_c(b, one));
        ctx.emit_at(kBoolC + col, bool_c(c, one));
        const F bc = b * c, sm = b + c;
        if (h) { b_hi = b_hi.dbl() + b; c_hi = c_hi.dbl() + c; }
        else { b_lo = b_lo.dbl() + b; c_lo = c_lo.dbl() + c; }
        ax[h] = ax[h].dbl() + (sm - bc.dbl());
        ao[h] = ao[h].dbl() + (sm - bc);
        aa[h] = aa[h].dbl() + bc;
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const F al = h ? a_hi : a_lo;
      ctx.emit_at(kBitwise + 0 + h, OPF(XOR) * (al - ax[h]));
      ctx.emit_at(kBitwise + 2 + h, OPF(OR) * (al - ao[h]));
      ctx.emit_at(kBitwise + 4 + h, OPF(AND) * (al - aa[h]));
    }
    ctx.note_limbs(0, b_lo, b_hi);
    ctx.note_limbs(1, c_lo, c_hi);
    const F k0 = L(C_K0), k1 = L(C_K1);
    const F immc = L(C_IS_REAL) - L(C_USE2);
    ctx.emit_at(kImm + 0, immc * (c_lo - L(C_IMM_LO)));
    ctx.emit_at(kImm + 1, immc * (c_hi - L(C_IMM_HI)));
    ctx.emit_at(kAddSub + 0, OPF(ADD) * (b_lo + c_lo - (a_lo + k65536 * k0)));
    ctx.emit_at(kAddSub + 1, OPF(ADD) * (b_hi + c_hi + k0 - (a_hi + k65536 * k1)));
    ctx.emit_at(kAddSub + 2, OPF(SUB) * (a_lo + c_lo - (b_lo + k65536 * k0)));
    ctx.emit_at(kAddSub + 3, OPF(SUB) * (a_hi + c_hi + k0 - (b_hi + k65536 * k1)));
    const F slt = OPF(SLT) + OPF(SLTU);
    ctx.emit_at(kCmp + 4, slt * (a_lo - k1));
    ctx.emit_at(kCmp + 5, slt * a_hi);
    const F np = L(C_NEXT_PC), tgt = L(C_TGT);
    ctx.emit_at(kNextPc + 2, OPF(JAL) * (a_lo - c_lo));
    ctx.emit_at(kNextPc + 3, OPF(JAL) * (a_hi - c_hi));
    ctx.emit_at(kNextPc + 4, OPF(JALR) * (a_lo + k65536 * a_hi - tgt));
    ctx.emit_at(kNextPc + 12, OPF(KECCAK) * (np - (b_lo + k65536 * b_hi)));
    const F code = ZKSP_K(0x02) * L(C_SC + SC_WRITE) + ZKSP_K(0x10) * L(C_SC + SC_COMMIT) + ZKSP_K(0x1a) * L(C_SC + SC_DEFER) +
                   ZKSP_K(0xf0) * L(C_SC + SC_HINT_LEN) + ZKSP_K(0xf1) * L(C_SC + SC_HINT_READ);
    ctx.emit_at(kEcall + 0, OPF(ECALL) * (b_lo - code));
    ctx.emit_at(kEcall + 1, OPF(ECALL) * b_hi);
    const F same = OPF(ECALL) - L(C_SC + SC_HINT_LEN);
    ctx.emit_at(kEcall + 2, same * (a_lo - b_lo));
    ctx.emit_at(kEcall + 3, same * (a_hi - b_hi));
  }
  if (TASK == 2) {
    F a_lo, a_hi, c_lo, c_hi;
    a_lo = L(C_A);
    a_hi = L(C_A + 1);
    limbs_of_block<F>(ctx, C_C, &c_lo, &c_hi);
    F samt = L(C_C + 4);
#pragma unroll
    for (int i = 3; i >= 0; --i) samt = samt.dbl() + L(C_C + i);
    const F c31 = L(C_C + 31);
    // X: booleans, limbs, one-hot sums.  Shifts through the prefix values P_n = sum_{i<n} 2^i b_i of B:
    //   sll  lo = sum_{k<16} x_k 2^k P_{16-k}               hi = sum_k x_k 2^(k-16) (P_{32-k} - P_{max(16-k,0)})
    //   srl  lo = sum_k x_k 2^-k (P_{min(16+k,32)} - P_k)    hi = sum_{k<16} x_k 2^-(16+k) (P_32 - P_{16+k})
    //   sra  = srl + b_31 sum_k x_k * (the 1-bits shifted in)
    // (the polynomials sum_j 2^j sum_k x_k b_{j -+ k} regrouped: 200 products instead of 2 100).  One loop over k
    // carries the four prefix values and the powers of two by recurrence, so nothing is indexed out of an array.
    F p16 = zero, p32 = zero, b31 = zero;
    {
      F pw = one;
      for (int i = 0; i < 32; ++i) {
        const F bi = L(C_B + i);
        ctx.stash(i, bi);
        p32 = p32 + pw * bi;
        if (i == 15) p16 = p32;
        if (i == 31) b31 = bi;
        pw = pw.dbl();
      }
    }
    const F inv2 = ctx.k(cmonty(inv_pow2_mod(1))), inv2_16 = ctx.k(cmonty(inv_pow2_mod(16)));
    const F b_lo = p16, b_hi = (p32 - p16) * inv2_16;
    F sum = zero, idx = zero, x_lo = zero, x_hi = zero, x0 = zero, x1 = zero;
    F sll_lo = zero, sll_hi = zero, srl_lo = zero, srl_hi = zero, fill_lo = zero, fill_hi = zero;
    {
      F pa = zero, pb = p16, pc = p16, pd = p32;        // P_k, P_min(16+k,32), P_max(16-k,0), P_(32-k)
      F pw = one, ipw = one, kf = zero;                  // 2^k, 2^-k, k
      F d15 = ctx.k(cmonty(pow2_mod(15))), d31 = ctx.k(cmonty(pow2_mod(31)));  // 2^(15-k), 2^(31-k)
      for (int k = 0; k < 32; ++k) {
        const F xk = L(C_X + k);
        ctx.emit_at(kBoolX + k, bool_c(xk, one));
        sum = sum + xk;
        idx = idx + kf * xk;
        if (k == 0) x0 = xk;
        if (k == 1) x1 = xk;
        if (k < 16) {
          x_lo = x_lo + pw * xk;
          sll_lo = sll_lo + xk * (pw * pc);
          srl_hi = srl_hi + xk * ((p32 - pb) * (ipw * inv2_16));
        } else {
          x_hi = x_hi + (pw * inv2_16) * xk;
        }
        sll_hi = sll_hi + xk * ((pd - pc) * (pw * inv2_16));
        srl_lo = srl_lo + xk * ((pb - pa) * ipw);
        if (k >= 17) fill_lo = fill_lo + xk * (k65536 - d31.dbl());
        if (k >= 16) fill_hi = fill_hi + xk * ZKSP_K(65535);
        else if (k >= 1) fill_hi = fill_hi + xk * (k65536 - d15.dbl());
        // step the recurrences to k + 1
        pa = pa + pw * ctx.stashed(k);
        if (k < 16) {
          pb = pb + (pw * k65536) * ctx.stashed(16 + k);
          pc = pc - d15 * ctx.stashed(15 - k);
          d15 = d15 * inv2;
        }
        pd = pd - d31 * ctx.stashed(31 - k);
        d31 = d31 * inv2;
        pw = pw.dbl();
        ipw = ipw * inv2;
        kf = kf + one;
      }
    }
    const F sh = OPF(SLL) + OPF(SRL) + OPF(SRA);
    ctx.emit_at(kShift + 0, sh * (sum - one));
    ctx.emit_at(kShift + 1, sh * (idx - samt));
    ctx.emit_at(kShift + 2, OPF(SLL) * (a_lo - sll_lo));
    ctx.emit_at(kShift + 3, OPF(SLL) * (a_hi - sll_hi));
    ctx.emit_at(kShift + 4, OPF(SRL) * (a_lo - srl_lo));
    ctx.emit_at(kShift + 5, OPF(SRL) * (a_hi - srl_hi));
    ctx.emit_at(kShift + 6, OPF(SRA) * (a_lo - (srl_lo + b31 * fill_lo)));
    ctx.emit_at(kShift + 7, OPF(SRA) * (a_hi - (srl_hi + b31 * fill_hi)));
    const F k0 = L(C_K0), k1 = L(C_K1), k2 = L(C_K2), k3 = L(C_K3), eq = L(C_EQ);
    {
      const F sgn = OPF(SLT) + OPF(BLT) + OPF(BGE);
      const F cmp = OPF(SLT) + OPF(SLTU) + OPF(BEQ) + OPF(BNE) + OPF(BLT) + OPF(BGE) + OPF(BLTU) + OPF(BGEU);
      ctx.emit_at(kCmp + 0, cmp * (b_lo - c_lo + k65536 * k0 - x_lo));
      ctx.emit_at(kCmp + 1, cmp * (b_hi - c_hi - k0 + k65536 * k1 - x_hi) + k65536 * (sgn * (c31 - b31)));
      const F bq = OPF(BEQ) + OPF(BNE), z = x_lo + x_hi;
      ctx.emit_at(kCmp + 2, bq * (z * L(C_INV) - one + eq));
      ctx.emit_at(kCmp + 3, bq * (z * eq));
    }
    {
      const F pc4 = L(C_PC) + ZKSP_K(4), np = L(C_NEXT_PC), tgt = L(C_TGT);
      const F def = L(C_IS_REAL) - OPF(JAL) - OPF(JALR) - OPF(BEQ) - OPF(BNE) - OPF(BLT) - OPF(BGE) - OPF(BLTU) - OPF(BGEU) - OPF(KECCAK);
      ctx.emit_at(kNextPc + 0, def * (np - pc4));
      ctx.emit_at(kNextPc + 1, OPF(JAL) * (np - tgt));
      ctx.emit_at(kNextPc + 5, OPF(JALR) * (np - (x_lo + k65536 * x_hi - x0)));
      const F d = tgt - pc4, base = np - pc4;
      ctx.emit_at(kNextPc + 6, OPF(BEQ) * (base - eq * d));
      ctx.emit_at(kNextPc + 7, OPF(BNE) * (base - (one - eq) * d));
      ctx.emit_at(kNextPc + 8, OPF(BLT) * (base - k1 * d));
      ctx.emit_at(kNextPc + 9, OPF(BGE) * (base - (one - k1) * d));
      ctx.emit_at(kNextPc + 10, OPF(BLTU) * (base - k1 * d));
      ctx.emit_at(kNextPc + 11, OPF(BGEU) * (base - (one - k1) * d));
    }
    const F loads = OPF(LB) + OPF(LH) + OPF(LW) + OPF(LBU) + OPF(LHU), stores = OPF(SB) + OPF(SH) + OPF(SW);
    const F ad = loads + stores + OPF(JALR);
    ctx.emit_at(kAddr + 0, ad * (b_lo + L(C_IMM_LO) - (x_lo + k65536 * k2)));
    ctx.emit_at(kAddr + 1, ad * (b_hi + L(C_IMM_HI) + k2 - (x_hi + k65536 * k3)));
    const F o0 = L(C_O0), o1 = L(C_O1), o2 = L(C_O2), o3 = L(C_O3), ls = loads + stores;
    ctx.emit_at(kOff + 0, ls * (o0 + o1 + (o2 + o3) - one));
    ctx.emit_at(kOff + 1, ls * (o1 + o2.dbl() + ZKSP_K(3) * o3 - (x0 + x1.dbl())));
    ctx.emit_at(kOff + 2, OPF(ECALL) * (o0 - one));
    ctx.emit_at(kOff + 3, OPF(ECALL) * (o1 + o2 + o3));
    ctx.emit_at(kOff + 4, OPF(ECALL) * (x_lo - ZKSP_K(11)));
    ctx.emit_at(kOff + 5, OPF(ECALL) * x_hi);
    ctx.note_limbs(3, x_lo, x_hi);
  }
  if (TASK == 3) {
    F a_lo, a_hi, c_lo, c_hi;
    a_lo = L(C_A);
    a_hi = L(C_A + 1);
    limbs_of_block<F>(ctx, C_C, &c_lo, &c_hi);
    F cb = L(C_C + 7);
#pragma unroll
    for (int i = 6; i >= 0; --i) cb = cb.dbl() + L(C_C + i);
    F m[32];
    load_bits(ctx, C_M, m);
#pragma unroll
    for (int i = 0; i < 32; ++i) ctx.emit_at(kBoolM + i, bool_c(m[i], one));
    const F m_lo = limb16(m, 0), m_hi = limb16(m, 1);
    ctx.note_limbs(2, m_lo, m_hi);
    const F mb[4] = {byte8(m, 0), byte8(m, 1), byte8(m, 2), byte8(m, 3)};
    const F mv_lo = L(C_MV_LO), mv_hi = L(C_MV_HI);
    const F k65535 = ZKSP_K(65535), k256 = ZKSP_K(256);
    const F o0 = L(C_O0), o1 = L(C_O1), o2 = L(C_O2), o3 = L(C_O3);
    const F oo[4] = {o0, o1, o2, o3};
    ctx.emit_at(kLoadStore + 0, OPF(LW) * (o0 - one));
    ctx.emit_at(kLoadStore + 1, OPF(LW) * (a_lo - m_lo));
    ctx.emit_at(kLoadStore + 2, OPF(LW) * (a_hi - m_hi));
    const F hv = o0 * m_lo + o2 * m_hi, hs = o0 * m[15] + o2 * m[31];
    ctx.emit_at(kLoadStore + 3, OPF(LHU) * (o1 + o3));
    ctx.emit_at(kLoadStore + 4, OPF(LHU) * (a_lo - hv));
    ctx.emit_at(kLoadStore + 5, OPF(LHU) * a_hi);
    ctx.emit_at(kLoadStore + 6, OPF(LH) * (o1 + o3));
    ctx.emit_at(kLoadStore + 7, OPF(LH) * (a_lo - hv));
    ctx.emit_at(kLoadStore + 8, OPF(LH) * (a_hi - k65535 * hs));
    F bv = zero, bs = zero;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      bv = bv + oo[p] * mb[p];
      bs = bs + oo[p] * m[8 * p + 7];
    }
    ctx.emit_at(kLoadStore + 9, OPF(LBU) * (a_lo - bv));
    ctx.emit_at(kLoadStore + 10, OPF(LBU) * a_hi);
    ctx.emit_at(kLoadStore + 11, OPF(LB) * (a_lo - (bv + ZKSP_K(0xff00) * bs)));
    ctx.emit_at(kLoadStore + 12, OPF(LB) * (a_hi - k65535 * bs));
    const F loads = OPF(LB) + OPF(LH) + OPF(LW) + OPF(LBU) + OPF(LHU);
    const F keep = loads + OPF(ECALL);
    ctx.emit_at(kLoadStore + 13, keep * (mv_lo - m_lo));
    ctx.emit_at(kLoadStore + 14, keep * (mv_hi - m_hi));
    ctx.emit_at(kLoadStore + 15, OPF(SW) * (o0 - one));
    ctx.emit_at(kLoadStore + 16, OPF(SW) * (mv_lo - c_lo));
    ctx.emit_at(kLoadStore + 17, OPF(SW) * (mv_hi - c_hi));
    ctx.emit_at(kLoadStore + 18, OPF(SH) * (o1 + o3));
    ctx.emit_at(kLoadStore + 19, OPF(SH) * (mv_lo - m_lo - o0 * (c_lo - m_lo)));
    ctx.emit_at(kLoadStore + 20, OPF(SH) * (mv_hi - m_hi - o2 * (c_lo - m_hi)));
    ctx.emit_at(kLoadStore + 21, OPF(SB) * (mv_lo - m_lo - (o0 * (cb - mb[0]) + k256 * (o1 * (cb - mb[1])))));
    ctx.emit_at(kLoadStore + 22, OPF(SB) * (mv_hi - m_hi - (o2 * (cb - mb[2]) + k256 * (o3 * (cb - mb[3])))));
  }
#undef OPF
}

template <class Ctx>
ZKSP_HD void eval_cpu(Ctx& ctx) {
  eval_cpu_task<0>(ctx);
  eval_cpu_task<1>(ctx);
  eval_cpu_task<2>(ctx);
  eval_cpu_task<3>(ctx);
  ctx.set_count(266);
}
constexpr int kCpuConstraints = 266;

template <class Ctx>
ZKSP_HD void eval_kmem(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1), is_first = ctx.is_first(), is_trans = ctx.is_trans();
  ctx.emit(bool_c(L(KM_IS_REAL), one));
  ctx.emit(bool_c(L(KM_ISF), one));
  ctx.emit(bool_c(L(KM_ISL), one));
  ctx.emit(L(KM_CALL) - L(KM_ISF) * L(KM_IS_REAL));
  ctx.emit(is_first * L(KM_IDX));
  ctx.emit(is_first * (L(KM_ISF) - one));
  const F nl = one - L(KM_ISL);
  ctx.emit(is_trans * (ctx.next(KM_IDX) - (L(KM_IDX) + one) * nl));
  ctx.emit(L(KM_ISL) * (L(KM_IDX) - ZKSP_K(49)));
  ctx.emit(is_trans * (ctx.next(KM_ISF) - L(KM_ISL)));
  ctx.emit(is_trans * nl * (ctx.next(KM_IS_REAL) - L(KM_IS_REAL)));
  ctx.emit(is_trans * ctx.next(KM_IS_REAL) * (one - L(KM_IS_REAL)));
  ctx.emit(is_trans * nl * (ctx.next(KM_TS) - L(KM_TS)));
  ctx.emit(is_trans * nl * (ctx.next(KM_PTR_LO) - L(KM_PTR_LO)));
  ctx.emit(is_trans * nl * (ctx.next(KM_PTR_HI) - L(KM_PTR_HI)));
  ctx.emit(L(KM_IS_REAL) * (L(KM_ADDR) - (L(KM_PTR_LO) + ZKSP_K(65536) * L(KM_PTR_HI) + ZKSP_K(4) * L(KM_IDX))));
  ctx.emit(L(KM_IS_REAL) * (L(KM_TS) + one - L(KM_PTS) - (L(KM_D) + ZKSP_K(1u << kTsLimbBits) * L(KM_D + 1))));
}
constexpr int kKmemConstraints = 16;

template <class Ctx>
ZKSP_HD void eval_memfinal(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1);
  ctx.emit(bool_c(L(MF_IS_REAL), one));
  ctx.emit(bool_c(L(MF_IS_INIT), one));
  for (int i = 0; i < 64; ++i) ctx.emit(bool_c(L(MF_DIFF + i), one));  // DIFF, INIT
  ctx.emit(L(MF_IS_INIT) * (one - L(MF_IS_REAL)));
  const F tn = ctx.is_trans() * ctx.next(MF_IS_REAL);
  ctx.emit(tn * (one - L(MF_IS_REAL)));
  ctx.emit(tn * (ctx.next(MF_ADDR) - L(MF_ADDR) - one - bits_val<F>(ctx, MF_DIFF, 32)));
}
constexpr int kMemFinalConstraints = 69;

template <class Ctx>
ZKSP_HD void eval_mul(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1);
  ctx.emit(bool_c(L(MU_IS_REAL), one));
  ctx.emit(bool_c(L(MU_HI), one));
  for (int i = 0; i < 32 + 32 + 64 + 31; ++i) ctx.emit(bool_c(L(MU_B + i), one));
  ctx.emit(L(MU_HI) * (one - L(MU_IS_REAL)));
  F b[4], c[4], sk[7];
  for (int i = 0; i < 4; ++i) {
    b[i] = byte_of<F>(ctx, MU_B, i);
    c[i] = byte_of<F>(ctx, MU_C, i);
  }
  for (int k = 0; k < 7; ++k) sk[k] = one - one;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) sk[i + j] = sk[i + j] + b[i] * c[j];
  const F q0 = bits_val<F>(ctx, MU_Q0, 10), q1 = bits_val<F>(ctx, MU_Q1, 11), q2 = bits_val<F>(ctx, MU_Q2, 10);
  const F k256 = ZKSP_K(256), k65536 = ZKSP_K(65536);
  ctx.emit(sk[0] + k256 * sk[1] - (limb_of<F>(ctx, MU_P, 0) + k65536 * q0));
  ctx.emit(sk[2] + k256 * sk[3] + q0 - (limb_of<F>(ctx, MU_P, 1) + k65536 * q1));
  ctx.emit(sk[4] + k256 * sk[5] + q1 - (limb_of<F>(ctx, MU_P, 2) + k65536 * q2));
  ctx.emit(sk[6] + q2 - limb_of<F>(ctx, MU_P, 3));
}
constexpr int kMulConstraints = 166;

template <class Ctx>
ZKSP_HD void eval_image(Ctx& ctx) {
  ctx.emit(bool_c(L(0), ctx.k(kR1)));
}
// the keccak chip's extra constraint after p3-keccak-air's 3182: the call time is constant inside
// a permutation's 24 rows
template <class Ctx>
ZKSP_HD void eval_keccak_ts(Ctx& ctx) {
  ctx.emit(ctx.is_trans() * (ctx.k(kR1) - L(ka::kFlags + 23)) * (ctx.next(KC_TS) - L(KC_TS)));
}
constexpr int kKeccakConstraints = ka::kNumConstraints + 1;
#undef L

ZKSP_HD constexpr int num_constraints(int chip) {
  return is_cpu_chip(chip) ? kCpuConstraints : chip == kKeccak ? kKeccakConstraints : chip == kKmem ? kKmemConstraints
       : chip == kMemFinal ? kMemFinalConstraints : chip == kImage ? 1 : chip == kProgram ? 0 : chip == kMul ? kMulConstraints : 0;
}

}  // namespace mach
}  // namespace zksp
