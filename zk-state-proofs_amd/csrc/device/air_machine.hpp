// Chips of the machine proof, format v6: column layouts and base-field
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
//
// The CPU row holds 16-bit limbs and no bits (since format v6).  It adds, subtracts, tests equality and unsigned order, moves words
// and forms addresses; xor / or / and go to the bitwise chip (bytes, looked up in the table chip's byte-operation
// columns), shifts and signed less-than to the ALU chip (bits), sub-word loads and stores to the sub-word chip (bytes),
// mul / mulhu to the multiplier, ecalls to the ecall chip, one row per such instruction.  Every row is an instruction (after HALT:
// the padding instruction the Program table ends with), so the Program lookup vouches for every decoded field.
// Range discipline: every producer of a memory-bus tuple guarantees canonical limbs (table lookups or bits), and
// addresses / jump targets stay below 0x78000000 < p, so no bus compares two 32-bit values that alias mod p.
#pragma once
#include "air_keccak.hpp"
#include "poseidon2.hpp"

namespace zksp {
namespace mach {

// ALU, sub-word and bitwise rows are each split over two instances of one AIR: the first has the largest power of two of
// rows strictly below the count, the second the rest (a power of two again).  The CPU rows (format v13) are spread over
// kNumCpuInst instances of one height, 2^ceil(log2(cycles / kNumCpuInst)): as many as the cycles need, the others at the
// minimum height (all padding): 391 400 cycles take 6 x 2^16 rows - the cells of 2^18 + 2^17, in rows six times as wide,
// so the trees, the quotient and the FRI domain have a quarter of the height and the per-row costs of a commitment are
// shared by six instances.
enum Chip { kCpu = 0, kKeccak, kKmem, kMemFinal, kImage, kProgram, kMul, kTable, kCpu2, kAlu, kAlu2, kSub, kSub2, kBw, kBw2, kP2, kEcall,
            kCpu3, kCpu4, kCpu5, kCpu6, kCpu7, kCpu8, kQr, kDiv, kTr, kHint, kNumChips };
constexpr int kNumCpuInst = 8;
// CPU instance i <-> chip (the first two keep their old places in the proof order)
ZKSP_HD constexpr int cpu_chip(int i) { return i == 0 ? kCpu : i == 1 ? kCpu2 : kCpu3 + (i - 2); }
ZKSP_HD constexpr int cpu_instance(int chip) { return chip == kCpu ? 0 : chip == kCpu2 ? 1 : chip >= kCpu3 && chip <= kCpu8 ? chip - kCpu3 + 2 : -1; }
// public scalars of a CPU instance: pc and time of its first row, whether another instance continues it, the pc
// that one starts at (the hand-over pc: a proof-header word the transcript absorbs), the padding pc (verifying key)
enum CpuPub { kPubStartPc = 0, kPubStartTs, kPubHasSucc, kPubEndPc, kPubPadPc, kNumCpuPub };
ZKSP_HD constexpr bool is_cpu_chip(int chip) { return cpu_instance(chip) >= 0; }
ZKSP_HD constexpr bool is_alu_chip(int chip) { return chip == kAlu || chip == kAlu2; }
ZKSP_HD constexpr bool is_sub_chip(int chip) { return chip == kSub || chip == kSub2; }
ZKSP_HD constexpr bool is_bw_chip(int chip) { return chip == kBw || chip == kBw2; }

// opcodes: Program table column CODE and the op element of the ALU / sub-word bus tuples
enum Op {
  ADD = 1, SUB, XOR, OR, AND, SLL, SRL, SRA, SLT, SLTU, JAL, JALR, BEQ, BNE, BLT, BGE, BLTU, BGEU, LB, LH, LW, LBU, LHU,
  SB, SH, SW, MUL, MULHU, ECALL, KECCAK, MULH, MULHSU, DIV, DIVU, REM, REMU
};
ZKSP_HD constexpr bool is_divrem(uint32_t op) { return op >= DIV && op <= REMU; }
ZKSP_HD constexpr bool is_mul_family(uint32_t op) { return op == MUL || op == MULHU || op == MULH || op == MULHSU; }
// instruction classes: one selector column each in the CPU row; Program table column CLS
enum Cls { CL_ADD = 1, CL_SUB, CL_ALU, CL_JAL, CL_JALR, CL_BEQ, CL_BNE, CL_BLT, CL_BGE, CL_LW, CL_SW, CL_LDS, CL_STS, CL_ECALL, CL_KECCAK };
constexpr int kNumCls = 15;
ZKSP_HD constexpr int class_of(uint32_t op) {
  return op == ADD ? CL_ADD : op == SUB ? CL_SUB : (op >= XOR && op <= SLTU) || is_mul_family(op) || is_divrem(op) ? CL_ALU
       : op == JAL ? CL_JAL : op == JALR ? CL_JALR : op == BEQ ? CL_BEQ : op == BNE ? CL_BNE
       : (op == BLT || op == BLTU) ? CL_BLT : (op == BGE || op == BGEU) ? CL_BGE : op == LW ? CL_LW : op == SW ? CL_SW
       : (op == LB || op == LH || op == LBU || op == LHU) ? CL_LDS : (op == SB || op == SH) ? CL_STS
       : op == ECALL ? CL_ECALL : op == KECCAK ? CL_KECCAK : 0;
}
// the op a row of that instruction puts on the ALU / sub-word bus (0: none)
ZKSP_HD constexpr uint32_t code_of(uint32_t op) {
  return ((op >= XOR && op <= SLTU) || is_mul_family(op) || is_divrem(op) || op == LB || op == LH || op == LBU || op == LHU ||
          op == SB || op == SH) ? op
       : (op == BLT || op == BGE) ? (uint32_t)SLT : (op == BLTU || op == BGEU) ? (uint32_t)SLTU : 0u;
}
// sltu, bltu, bgeu: the unsigned comparison the CPU row does itself (Program column UC)
ZKSP_HD constexpr bool ucmp_of(uint32_t op) { return op == SLTU || op == BLTU || op == BGEU; }
// does a cycle of this op occupy a row of the ALU chip (0) / the sub-word chip (1) / the bitwise chip (2) / the ecall
// chip (3) / the divider chip (4)?  (-1: none)
ZKSP_HD constexpr int event_kind(uint32_t op) {
  return (code_of(op) >= SLL && code_of(op) <= SLT) ? 0
       : (op == LB || op == LH || op == LBU || op == LHU || op == SB || op == SH) ? 1
       : (op >= XOR && op <= AND) ? 2 : op == ECALL ? 3 : is_divrem(op) ? 4 : -1;
}

// ---- CPU chip ----
constexpr int C_PC = 0, C_TS = 1, C_NEXT_PC = 2, C_SEL = 3, C_CODE = C_SEL + kNumCls, C_UC = C_CODE + 1, C_WR = C_CODE + 2,
              C_USE2 = C_CODE + 3, C_RD = C_CODE + 4, C_RS1 = C_CODE + 5, C_RS2 = C_CODE + 6, C_IMM_LO = C_CODE + 7,
              C_IMM_HI = C_CODE + 8, C_TGT_LO = C_CODE + 9, C_TGT_HI = C_CODE + 10, C_A = C_CODE + 11, C_B = C_A + 2, C_C = C_B + 2,
              C_X = C_C + 2, C_K0 = C_X + 2, C_K1 = C_K0 + 1, C_O1 = C_K0 + 2, C_O2 = C_O1 + 1, C_O3 = C_O1 + 2,
              C_ADDR2 = C_O1 + 3, C_ADDR3 = C_ADDR2 + 1, C_W_PLO = C_ADDR3 + 1, C_W_PHI = C_W_PLO + 1, C_GAP = C_W_PLO + 2,
              kCpuWidth = C_GAP + 6;
static_assert(kCpuWidth == 52, "CPU chip layout: three accesses per row, seven Poseidon2 absorptions");
// ---- ecall chip: one row per ecall (the CPU row of an ecall moves t0 only and hands the rest over on the ECALL bus) ----
enum { SC_HALT = 0, SC_WRITE, SC_COMMIT, SC_DEFER, SC_HINT_LEN, SC_HINT_READ };
//      (format v16: a HINT_READ of len bytes at ptr announces its NW = ceil(len / 4) words to the hint chip: 4 NW = len + P1 + 2 P2)
constexpr int EC_IS_REAL = 0, EC_SC = 1, EC_TS = EC_SC + 6, EC_PC = EC_TS + 1, EC_NP = EC_TS + 2, EC_B_LO = EC_TS + 3, EC_A_LO = EC_TS + 4,
              EC_A_HI = EC_TS + 5, EC_C_LO = EC_TS + 6, EC_C_HI = EC_TS + 7, EC_M_LO = EC_TS + 8, EC_M_HI = EC_TS + 9, EC_GAP = EC_TS + 10,
              EC_NW = EC_GAP + 4, EC_P1 = EC_NW + 1, EC_P2 = EC_NW + 2, kEcallWidth = EC_NW + 3;
static_assert(kEcallWidth == 24, "ecall chip layout");
ZKSP_HD constexpr int selc(int cls) { return C_SEL + cls - 1; }

// ---- keccak chip: p3-keccak-air's columns + the call time ----
constexpr int KC_TS = ka::kWidth, kKeccakWidth = ka::kWidth + 1;
// ---- keccak-memory chip ----
constexpr int KM_IS_REAL = 0, KM_TS = 1, KM_PTR_LO = 2, KM_PTR_HI = 3, KM_IDX = 4, KM_ISF = 5, KM_ISL = 6, KM_CALL = 7,
              KM_ADDR = 8, KM_OLD_LO = 9, KM_OLD_HI = 10, KM_NEW_LO = 11, KM_NEW_HI = 12, KM_PTS = 13, KM_GL = 14, KM_GH = 15,
              kKmemWidth = 16;
// ---- memory boundary chip: EVERY image address and every other touched address once, strictly increasing ----
constexpr int MF_IS_REAL = 0, MF_LO = 1, MF_HI = 2, MF_IS_INIT = 3, MF_INIT_LO = 4, MF_INIT_HI = 5, MF_FIN_LO = 6, MF_FIN_HI = 7,
              MF_FIN_TS = 8, MF_D_LO = 9, MF_D_HI = 10, MF_BW = 11,
              // format v16: IS_INIT - a hinted word: its initial value comes from the hint chip (over the IMG bus, as an image
              // word's comes from the image chip); IS_ZERO - any other address outside the image: it starts as zero
              MF_IS_ZERO = 12, kMemFinalWidth = 13;
// ---- hint chip (format v16): one row per word of every HINT_READ, in address order within a read: the read's first row takes
//      (pointer, number of words) from the ecall chip; a row the run touches (USED) puts (address, initial value) on the IMG
//      bus, where the memory boundary chip takes a hinted word's initial value from - so memory outside the image starts with
//      the prover's choice ONLY where a HINT_READ put input, and with zero everywhere else ----
constexpr int HN_IS_REAL = 0, HN_FIRST = 1, HN_LAST = 2, HN_ADDR = 3, HN_CNT = 4, HN_LO = 5, HN_HI = 6, HN_USED = 7, kHintWidth = 8;
// ---- image / program chips: preprocessed columns, one main column ----
constexpr int IMG_P_ADDR = 0, IMG_P_LO = 1, IMG_P_HI = 2, IMG_P_REAL = 3, kImagePrepWidth = 4, kImageWidth = 1;
constexpr int PR_PC = 0, PR_CLS = 1, PR_CODE = 2, PR_UC = 3, PR_WR = 4, PR_USE2 = 5, PR_RD = 6, PR_RS1 = 7, PR_RS2 = 8, PR_IMM_LO = 9,
              PR_IMM_HI = 10, PR_TGT_LO = 11, PR_TGT_HI = 12, kProgramPrepWidth = 13, kProgramWidth = 1;
// ---- multiplier chip ----
//      (format v15: mulh / mulhsu as well - the signed high words R follow from the unsigned product's high word P_hi:
//      R + b31 * C + [mulh] c31 * B = P_hi + 2^32 k, limb by limb with carries K0, K1 in {0, 1, 2} as two bits each; R's limbs are
//      looked up in the range table)
constexpr int MU_IS_REAL = 0, MU_HI = 1, MU_B = 2, MU_C = MU_B + 32, MU_P = MU_C + 32, MU_Q0 = MU_P + 64, MU_Q1 = MU_Q0 + 10,
              MU_Q2 = MU_Q1 + 11, MU_SH = MU_Q2 + 10, MU_SHU = MU_SH + 1, MU_R = MU_SH + 2, MU_K0 = MU_R + 2, MU_K1 = MU_K0 + 2,
              kMulWidth = MU_K1 + 2;
static_assert(kMulWidth == 169, "multiplier chip layout");
// ---- divider chip (format v15): div divu rem remu, one row per instruction.  On absolute values |n| = |q| |d| + |r| with
//      |r| < |d|; the product comes from the multiplier chip over the ALU bus (low word PL, high word zero); signs: q has
//      sign(n) xor sign(d) unless it is zero, r the sign of n unless it is zero; a zero divisor gives q = 0xffffffff, r = n;
//      -2^31 / -1 gives q = -2^31, r = 0 by the same relations.  N, D the operands, A the result the CPU row gets ----
constexpr int DV_IS_REAL = 0, DV_F = 1, DV_N = 5, DV_D = 7, DV_A = 9, DV_SN = 11, DV_SD = 12, DV_NH = 13, DV_DH = 14, DV_AN = 15,
              DV_AD = 17, DV_AQ = 19, DV_AR = 21, DV_CN = 23, DV_CD = 24, DV_CQ = 25, DV_CR = 26, DV_Q = 27, DV_R = 29, DV_SQ = 31,
              DV_SR = 32, DV_XS = 33, DV_PL = 34, DV_K = 36, DV_E = 37, DV_BE = 39, DV_NZD = 40, DV_INVD = 41, DV_NZQ = 42,
              DV_INVQ = 43, DV_NZR = 44, DV_INVR = 45, kDivWidth = 46;
// ---- ALU chip: sll srl sra and the signed slt (slt, blt, bge) over bits ----
constexpr int AL_IS_REAL = 0, AL_SEL = 1, AL_A = AL_SEL + 4, AL_B = AL_A + 2, AL_C = AL_B + 32, AL_X = AL_C + 32, AL_K0 = AL_X + 32,
              AL_K1 = AL_K0 + 1, kAluWidth = AL_K1 + 1;
static_assert(kAluWidth == 105, "ALU chip layout");
// ---- bitwise chip: xor or and byte by byte, every (b, c, a) byte triple looked up in the table chip ----
constexpr int BW_IS_REAL = 0, BW_SEL = 1, BW_A = BW_SEL + 3, BW_B = BW_A + 4, BW_C = BW_B + 4, kBwWidth = BW_C + 4;
static_assert(kBwWidth == 16, "bitwise chip layout");
// ---- sub-word chip: lb lh lbu lhu sb sh ----
constexpr int SW_IS_REAL = 0, SW_SEL = 1, SW_O = SW_SEL + 6, SW_A = SW_O + 4, SW_MB = SW_A + 2, SW_CB = SW_MB + 4,
              SW_S = SW_CB + 2, SW_SELB = SW_S + 1, kSubWidth = SW_SELB + 1;
static_assert(kSubWidth == 21, "sub-word chip layout");
// ---- Poseidon2 chip: one width-16 permutation per row.  Six kinds of rows:
//   N   (stage 1) a node K of a heap of digests (root 1, children 2K and 2K + 1): consumes its children's digests from the
//       DIGEST bus, produces its own; the verifier supplies the leaves and takes the root;
//   SZ / SC  (stage 2a: the openings of a leaf proof) a sponge row: eight absorbed words over the capacity carried from the
//       row before (SC) or over the zero state (SZ: the first block of a hash);
//   PL / PR  a step of a Merkle path: the running digest (the output of the row before) is the left / right input, the
//       sibling the other, free; K' = 2 K + [right], M' = 2 M;
//   J   an injection of a mixed-height tree: the running digest on the left, on the right the hash of the shorter matrices'
//       row, consumed from the DIGEST bus where the sponge that made it (a segment of S rows elsewhere, labelled with this
//       row's T, K, M) put it; K' = K, M' = M + 1.
// A run (one opening) starts with a sponge over the zero state flagged NEW (K = 1, M = 0): K collects the position bits,
// M the levels an injection followed, T names the opening; its last row sends (T, 0, K, M, digest), which the verifier
// consumes with the root it knows - so position, shape and root of every opening are the verifier's.  Columns: the
// input state, and per S-box its cube and its seventh power. ----
constexpr int P2_IS_REAL = 0, P2_KL = 1, P2_KH = 2, P2_T = 3, P2_M = 4, P2_FN = 5, P2_SZ = 6, P2_SC = 7, P2_PL = 8, P2_PR = 9, P2_FJ = 10,
              P2_NEW = 11, P2_SND = 12, P2_FR = 13, P2_IN = 14, P2_EXT = P2_IN + 16, P2_INT = P2_EXT + 256,
              // format v16 (row f4, stage 2b): RE - the last row of a run (its digest is compared with the root the transcript
              // chip holds, its position goes to the query chip); SE - the last block of a matrix row's hash (the Horner sum of
              // the absorbed words goes to the query chip); RID - which root; SO - the Horner sum after this block; AP - the
              // powers alpha_f^1 .. alpha_f^8 (eight extension elements)
              P2_RE = P2_INT + 26, P2_SE = P2_RE + 1, P2_RID = P2_RE + 2, P2_SO = P2_RE + 3, P2_AP = P2_SO + 4, kP2Width = P2_AP + 32;
static_assert(kP2Width == 351, "Poseidon2 chip layout");
// records the rows are expanded from (32 words): flags, T, K, M, the 16 input words, RID, SO (4), alpha_f (4)
enum P2Kind { P2K_NONE = 0, P2K_NODE, P2K_SZ, P2K_SC, P2K_PL, P2K_PR, P2K_J };
constexpr uint32_t kP2RecWords = 32, kP2FlagNew = 16, kP2FlagSnd = 32, kP2FlagFri = 64, kP2FlagRe = 128, kP2FlagSe = 256;  // flags = kind | ...
constexpr int kP2RecRid = 20, kP2RecSo = 21, kP2RecAlpha = 25;
// tags of a leaf proof's openings: leaf l (its place among the leaves checked beside one run), query q, tree r (0 preprocessed,
// 1 main, 2 permutation, 3 quotient, 4 + k FRI layer k); ids of its commitment roots
constexpr uint32_t kLeafTagStride = 64, kLeafTagLeafStride = 1u << 18, kLeafMaxQueries = 4096;
ZKSP_HD constexpr uint32_t leaf_tag(uint32_t l, uint32_t q, uint32_t r) { return 1 + kLeafTagStride * q + kLeafTagLeafStride * l + r; }
ZKSP_HD constexpr uint32_t leaf_rid(uint32_t l, uint32_t r) { return 64 * l + r; }
// ---- query chip (row f4, stage 2b; it stands where stage 2a's fold chip stood): 31 rows per query of a checked leaf proof, one
//      per bit of the word the leaf's transcript drew for the query, from bit 30 down.  The bits are the canonical
//      decomposition of the word; bit lm of it is the coset, the bits below the position.  The rows of the bits lm - 1 .. 0
//      are the FRI layers 0 .. lm - 1: each receives its layer's opening (position, pair, challenge), folds, and - where a
//      height joins - forms the reduced opening of that height from the Horner sums of the opened rows. ----
constexpr int QR_IS_REAL = 0, QR_FIRST = 1, QR_LAST = 2, QR_LEAF = 3, QR_QL = 4, QR_J = 5, QR_BIT = 6, QR_ACC = 7, QR_EQ = 8, QR_F1 = 9,
              QR_F2 = 10, QR_F3 = 11, QR_CSR = 12, QR_FL = 13, QR_LAY = 14, QR_K = 15, QR_CS = 16, QR_POW = 17, QR_LOW = 18, QR_REV = 19,
              QR_PR0 = 20, QR_CNT0 = 21, QR_KEYJ = 22, QR_MJ = 23, QR_MT = 24, QR_P0A = 25, QR_KEY0 = 26, QR_M0 = 27, QR_MT0 = 28,
              QR_HASRO = 29, QR_HAS0 = 30, QR_OMI = 31, QR_MU = 32, QR_CSM = 33, QR_R = 34, QR_R2 = 35, QR_YT = 36, QR_YKI = 37,
              QR_GI = 38, QR_XINV = 39, QR_WH = 40,
              QR_BETA = 41, QR_LO = 45, QR_HI = 49, QR_E = 53, QR_F = 57, QR_RO = 61, QR_H = 65 /* 4 x 4 */, QR_AF = 81, QR_DL = 85,
              QR_D2 = 89, QR_D3 = 93, QR_D4 = 97, QR_G2 = 101, QR_ZETA = 105, QR_ZW = 109, QR_D0 = 113, QR_D1 = 117, QR_B1 = 121,
              QR_B2 = 125, kQrWidth = 129;
constexpr uint32_t kQrRecWords = 132;  // a row record is the row: kQrWidth canonical words (then padding)
// ---- transcript chip (row f4, stage 2b): one duplex of a checked leaf proof's Fiat-Shamir transcript per row - "absorb eight
//      words" or "squeeze" (since format v16 every phase of a transcript ends on a block boundary).  The verifier dictates
//      every row (what it absorbs, what its inputs and outputs are used for); the rows hand the commitment roots, the final
//      constant and the challenges zeta, alpha_f / delta and the FRI betas to the chips that check the queries, and every
//      query's index word to the query chip. ----
constexpr int TR_IS_REAL = 0, TR_LEAF = 1, TR_STEP = 2, TR_FIRST = 3, TR_ABS = 4, TR_UROOT = 5, TR_UZETA = 6, TR_UAF = 7, TR_UBETA = 8,
              TR_UFIN = 9, TR_UPOW = 10, TR_UQ = 11, TR_QM = 12 /* 8 */, TR_RIDK = 20, TR_QBASE = 21, TR_MROOT = 22, TR_MFIN = 23,
              TR_MZETA = 24, TR_MAF = 25, TR_MBETA = 26, TR_IN = 27, TR_EXT = TR_IN + 16, TR_INT = TR_EXT + 256, kTrWidth = TR_INT + 26;
static_assert(kTrWidth == 325, "transcript chip layout");
// row records (32 words): word 0 = the flags word of the verifier's tuple (UROOT + 2 UZETA + 4 UAF + 8 UBETA + 16 UFIN + 32 UPOW
// + 64 UQ + 128 * the QM bits) | FIRST << 16 | ABS << 17; leaf, step, RIDK, QBASE, the five multiplicities, the 16 input words
constexpr uint32_t kTrRecWords = 32, kTrRecFirst = 1u << 16, kTrRecAbs = 1u << 17;
// ---- table chip: 2^16 rows; preprocessed (x, y: the row index's bytes; na: index not a multiple of 4; nt: index above
//      kAddrHiMax or zero; x ^ y; x & y); main: multiplicities of range16 (kind 0), 4-aligned range16 (kind 1), high address limb
//      (kind 2), byte pair, and the byte operations xor / or / and ----
constexpr int TB_P_X = 0, TB_P_Y = 1, TB_P_NA = 2, TB_P_NT = 3, TB_P_XOR = 4, TB_P_AND = 5, kTablePrepWidth = 6, TB_M_R16 = 0,
              TB_M_AL = 1, TB_M_TOP = 2, TB_M_BY = 3, TB_M_XOR = 4, TB_M_OR = 5, TB_M_AND = 6, kTableWidth = 7, kTableLogH = 16;
// high limb of the largest address / jump target; the smallest is 1: a load, store or keccak state below 0x10000 has no
// table row, so no memory access can name a register (addresses 0 .. 31 on the memory bus)
constexpr uint32_t kAddrHiMax = 0x77FEu;
constexpr uint32_t kGenInv = 64944062u;  // 1 / kGen mod p (the LDE cosets' shift is the generator)

enum Bus { BUS_MEM = 1, BUS_PROG, BUS_KCALL, BUS_KIO, BUS_ALU, BUS_PUBC, BUS_PUBH, BUS_RANGE, BUS_BYTES, BUS_SUB, BUS_IMG, BUS_BYTEOP, BUS_DIGEST, BUS_ECALL, BUS_PAIR,
           // stage 2b: a run's position -> query chip; commitment roots; a matrix row's Horner sum; the transcript's blocks and
           // squeezes (verifier -> transcript chip); final constant; zeta; alpha_f, delta; FRI betas; the proof-of-work word
           // (-> verifier); query index words; per leaf constants and per height constants (verifier -> query chip)
           BUS_POS, BUS_ROOT, BUS_SEG, BUS_TBLK, BUS_TSQ, BUS_FINAL, BUS_ZETA, BUS_AF, BUS_BETA, BUS_POW, BUS_QIDX, BUS_LEAFK, BUS_BCONST,
           BUS_HINTR /* (pointer, words): ecall chip -> hint chip */ };

// Ctx interface:
//   using F;  F local(int col); F next(int col); F prep(int col) (preprocessed column of the row);
//   F is_first(); F is_trans(); F is_last(); F pub(int which)  (CpuPub);  const P2Consts* p2()  (eval_p2 only);
//   F k(uint32_t montgomery_word)  (a constant);  void emit(F v)  (appends the next constraint);
//   void emit_at(int index, F v);  void set_count(int n)  (index of the next emit());
//   void stash(int i, F v); F stashed(int i)  (eval_alu_task<1> only): 32 values parked by index and read back by
//     index (the shift constraints revisit B's bits in a loop the device compiler keeps rolled: the device parks
//     them in LDS instead of going back to HBM three more times per bit)
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

ZKSP_HD constexpr uint32_t pow2_mod(int n) { return (uint32_t)(((uint64_t)1 << n) % kP); }
ZKSP_HD constexpr uint32_t inv_pow2_mod(int n) {  // 2^-n mod p
  uint64_t r = 1;
  for (int i = 0; i < n; ++i) r = r * ((kP + 1) / 2) % kP;
  return (uint32_t)r;
}

#define L(c) ctx.local(c)

// ---- CPU chip: 69 constraints, emitted in order (three accesses per row) ----
template <class Ctx>
ZKSP_HD void eval_cpu(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1), zero = one - one;
  const F k65536 = ZKSP_K(65536);
#define S(cls) ctx.local(selc(cls))
  // booleans: class selectors, carries, byte offset (WR, USE2 are Program-table values)
  F selsum = zero;
#pragma unroll
  for (int k = 0; k < kNumCls; ++k) {
    const F v = L(C_SEL + k);
    ctx.emit(bool_c(v, one));
    selsum = selsum + v;
  }
  const F k0 = L(C_K0), k1 = L(C_K1);
  ctx.emit(bool_c(k0, one));
  ctx.emit(bool_c(k1, one));
  const F o1 = L(C_O1), o2 = L(C_O2), o3 = L(C_O3), osum = o1 + o2 + o3;
  ctx.emit(bool_c(o1, one)); ctx.emit(bool_c(o2, one)); ctx.emit(bool_c(o3, one));
  // row structure: exactly one class; the clock; the chain of pcs; the instance's first and last rows
  const F pc = L(C_PC), ts = L(C_TS), np = L(C_NEXT_PC), pad_pc = ctx.pub(kPubPadPc);
  ctx.emit(selsum - one);
  ctx.emit(ctx.is_first() * (pc - ctx.pub(kPubStartPc)));
  ctx.emit(ctx.is_first() * (ts - ctx.pub(kPubStartTs)));
  ctx.emit(ctx.is_trans() * (ctx.next(C_TS) - ts - ZKSP_K(4)));
  ctx.emit(ctx.is_trans() * (ctx.next(C_PC) - np));
  {
    const F succ = ctx.pub(kPubHasSucc);
    ctx.emit(ctx.is_last() * succ * (np - ctx.pub(kPubEndPc)));
    ctx.emit(ctx.is_last() * (one - succ) * (np - pad_pc));
  }
  const F a_lo = L(C_A), a_hi = L(C_A + 1), b_lo = L(C_B), b_hi = L(C_B + 1), c_lo = L(C_C), c_hi = L(C_C + 1);
  const F x_lo = L(C_X), x_hi = L(C_X + 1);
  const F imm_lo = L(C_IMM_LO), imm_hi = L(C_IMM_HI);
  const F loadw = S(CL_LW) + S(CL_LDS), storew = S(CL_SW) + S(CL_STS);
  // operand C is the immediate
  {
    const F immc = one - L(C_USE2) - loadw;
    ctx.emit(immc * (c_lo - imm_lo));
    ctx.emit(immc * (c_hi - imm_hi));
  }
  // the adder: X = B + C (add, jalr, lw, sub-word loads), X = B + imm (stores), X + C = B (sub)
  {
    const F addc = S(CL_ADD) + S(CL_JALR), addi = loadw + storew;
    ctx.emit(addc * (b_lo + c_lo - (x_lo + k65536 * k0)));
    ctx.emit(addc * (b_hi + c_hi + k0 - (x_hi + k65536 * k1)));
    ctx.emit(addi * (b_lo + imm_lo - (x_lo + k65536 * k0)));
    ctx.emit(addi * (b_hi + imm_hi + k0 - (x_hi + k65536 * k1)));
    ctx.emit(S(CL_SUB) * (x_lo + c_lo - (b_lo + k65536 * k0)));
    ctx.emit(S(CL_SUB) * (x_hi + c_hi + k0 - (b_hi + k65536 * k1)));
    // what the range lookups check is X: the sum / difference written (add, sub), the value an ecall leaves in t0
    // (HINT_LEN: prover-supplied), the return address of a keccak call
    const F cpa = S(CL_ADD) + S(CL_SUB) + S(CL_ECALL);
    ctx.emit(cpa * (a_lo - x_lo));
    ctx.emit(cpa * (a_hi - x_hi));
    ctx.emit(S(CL_KECCAK) * (x_lo - b_lo));
    ctx.emit(S(CL_KECCAK) * (x_hi - b_hi));
    // unsigned comparison in the row (sltu, bltu, bgeu): X = B - C + 2^32 K1 limb by limb, so K1 = [B < C] because both
    // limbs of X are looked up in the range table; the flag goes where the ALU chip's answer would
    const F uc = L(C_UC);
    ctx.emit(uc * (b_lo - c_lo + k65536 * k0 - x_lo));
    ctx.emit(uc * (b_hi - c_hi - k0 + k65536 * k1 - x_hi));
    ctx.emit(uc * (a_lo - k1));
    ctx.emit(uc * a_hi);
  }
  // byte offset and the word address
  const F off = o1 + o2.dbl() + ZKSP_K(3) * o3;
  const F xaddr = x_lo + k65536 * x_hi - off;
  {
    const F noff = S(CL_ADD) + S(CL_SUB) + S(CL_ECALL) + S(CL_KECCAK) + S(CL_LW) + S(CL_SW) + L(C_UC);
    ctx.emit(noff * osum);
    ctx.emit(bool_c(osum, one));  // at most one of the three offset flags
    ctx.emit(S(CL_JALR) * (o2 + o3));
    // the second access is rs2 or a load's word, the written location rd or a store's word
    ctx.emit(L(C_USE2) * (L(C_ADDR2) - L(C_RS2)));
    ctx.emit(loadw * (L(C_ADDR2) - xaddr));
    ctx.emit(L(C_WR) * (L(C_ADDR3) - L(C_RD)));
    ctx.emit(storew * (L(C_ADDR3) - xaddr));
  }
  // next pc
  {
    const F pc4 = pc + ZKSP_K(4), tgt = L(C_TGT_LO) + k65536 * L(C_TGT_HI);
    const F def = one - S(CL_JAL) - S(CL_JALR) - S(CL_BEQ) - S(CL_BNE) - S(CL_BLT) - S(CL_BGE) - S(CL_KECCAK) - S(CL_ECALL);
    ctx.emit(def * (np - pc4));
    ctx.emit(S(CL_JAL) * (np - tgt));
    ctx.emit(S(CL_JAL) * (a_lo - c_lo));
    ctx.emit(S(CL_JAL) * (a_hi - c_hi));
    ctx.emit(S(CL_JALR) * (a_lo - L(C_TGT_LO)));
    ctx.emit(S(CL_JALR) * (a_hi - L(C_TGT_HI)));
    ctx.emit(S(CL_JALR) * (np - xaddr));
    // beq / bne: a limb difference is zero or has the inverse X holds; the flag is "both limbs equal"
    const F bq = S(CL_BEQ) + S(CL_BNE), d_lo = b_lo - c_lo, d_hi = b_hi - c_hi;
    ctx.emit(bq * (d_lo * x_lo - one + k0));
    ctx.emit(bq * (d_lo * k0));
    ctx.emit(bq * (d_hi * x_hi - one + k1));
    ctx.emit(bq * (d_hi * k1));
    ctx.emit(bq * (a_lo - k0 * k1));
    ctx.emit((bq + S(CL_BLT) + S(CL_BGE)) * a_hi);
    // taken -> tgt, else pc + 4
    const F d = tgt - pc4, base = np - pc4;
    ctx.emit((S(CL_BEQ) + S(CL_BLT)) * (base - a_lo * d));
    ctx.emit((S(CL_BNE) + S(CL_BGE)) * (base - (one - a_lo) * d));
    ctx.emit(S(CL_KECCAK) * (np - (b_lo + k65536 * b_hi)));
    // (ecall: the ecall chip decides the next pc - the next instruction, or the padding instruction after HALT)
  }
  // word loads and stores: the register gets the word read (C), the memory the register's value (C)
  {
    const F mov = S(CL_LW) + S(CL_SW);
    ctx.emit(mov * (a_lo - c_lo));
    ctx.emit(mov * (a_hi - c_hi));
  }
  // ecall: the code in t0 is a 16-bit value (decoded, and the value left behind checked, by the ecall chip)
  ctx.emit(S(CL_ECALL) * b_hi);
  // previous access times are older by construction: a slot's previous time IS its time - 1 - difference (a linear form in
  // the memory-bus tuples), and the difference's low limb and high byte are looked up in the table chip
#undef S
}
constexpr int kCpuConstraints = 69 - 6 + 4 + 2;

// ---- ecall chip: the code in t0 decoded into six flags; t0 rewritten with itself except by HINT_LEN (whose answer the CPU
// row range-checks); the next pc: the next instruction, or the padding instruction after HALT ----
template <class Ctx>
ZKSP_HD void eval_ecall(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1), zero = one - one, real = L(EC_IS_REAL);
  ctx.emit(bool_c(real, one));
  F scsum = zero;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const F v = L(EC_SC + k);
    ctx.emit(bool_c(v, one));
    scsum = scsum + v;
  }
  const F code = ZKSP_K(0x02) * L(EC_SC + SC_WRITE) + ZKSP_K(0x10) * L(EC_SC + SC_COMMIT) + ZKSP_K(0x1a) * L(EC_SC + SC_DEFER) +
                 ZKSP_K(0xf0) * L(EC_SC + SC_HINT_LEN) + ZKSP_K(0xf1) * L(EC_SC + SC_HINT_READ);
  ctx.emit(scsum - real);
  ctx.emit(L(EC_B_LO) - code);
  const F same = real - L(EC_SC + SC_HINT_LEN);
  ctx.emit(same * (L(EC_A_LO) - L(EC_B_LO)));
  ctx.emit(same * L(EC_A_HI));
  const F pc4 = L(EC_PC) + ZKSP_K(4);
  ctx.emit(real * (L(EC_NP) - pc4) - L(EC_SC + SC_HALT) * (ctx.pub(kPubPadPc) - pc4));
  // COMMIT / COMMIT_DEFERRED: the word index in a0 is the whole register (the PUBC tuple carries its low limb only)
  ctx.emit((L(EC_SC + SC_COMMIT) + L(EC_SC + SC_DEFER)) * L(EC_C_HI));
  // HINT_READ: the number of words the read covers (NW is looked up as 16 bits: at most 2^18 bytes per read)
  const F hr = L(EC_SC + SC_HINT_READ), p1 = L(EC_P1), p2 = L(EC_P2);
  ctx.emit(bool_c(p1, one));
  ctx.emit(bool_c(p2, one));
  ctx.emit((p1 + p2) * (one - hr));
  ctx.emit((one - hr) * L(EC_NW));
  ctx.emit(hr * (ZKSP_K(4) * L(EC_NW) - p1 - p2.dbl() - L(EC_M_LO) - ZKSP_K(65536) * L(EC_M_HI)));
}
constexpr int kEcallConstraints = 18;

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
  ctx.emit(L(KM_IS_REAL) * (L(KM_TS) + one - L(KM_PTS) - (L(KM_GL) + ZKSP_K(65536) * L(KM_GH))));
}
constexpr int kKmemConstraints = 16;

template <class Ctx>
ZKSP_HD void eval_memfinal(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1);
  ctx.emit(bool_c(L(MF_IS_REAL), one));
  ctx.emit(bool_c(L(MF_IS_INIT), one));
  ctx.emit(bool_c(L(MF_BW), one));
  ctx.emit(L(MF_IS_INIT) * (one - L(MF_IS_REAL)));
  const F tn = ctx.is_trans() * ctx.next(MF_IS_REAL);
  ctx.emit(tn * (one - L(MF_IS_REAL)));
  // next address - address - 1 = D >= 0, limb by limb with a borrow: every term stays far below p, so this is a
  // statement about integers (all six limbs are looked up in the table chip)
  ctx.emit(tn * (ctx.next(MF_LO) - L(MF_LO) - one + ZKSP_K(65536) * L(MF_BW) - L(MF_D_LO)));
  ctx.emit(tn * (ctx.next(MF_HI) - L(MF_HI) - L(MF_BW) - L(MF_D_HI)));
  // an address outside the image is hinted (IS_INIT) or starts as zero (IS_ZERO)
  const F z = L(MF_IS_ZERO);
  ctx.emit(bool_c(z, one));
  ctx.emit(z * (one - L(MF_IS_REAL)));
  ctx.emit(z * L(MF_IS_INIT));
  ctx.emit(z * L(MF_INIT_LO));
  ctx.emit(z * L(MF_INIT_HI));
}
constexpr int kMemFinalConstraints = 12;

// ---- hint chip ----
template <class Ctx>
ZKSP_HD void eval_hint(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1), real = L(HN_IS_REAL), first = L(HN_FIRST), last = L(HN_LAST), used = L(HN_USED);
  ctx.emit(bool_c(real, one)); ctx.emit(bool_c(first, one)); ctx.emit(bool_c(last, one)); ctx.emit(bool_c(used, one));
  ctx.emit((first + last + used) * (one - real));
  ctx.emit(ctx.is_trans() * ctx.next(HN_IS_REAL) * (one - real));  // the real rows are a prefix
  ctx.emit(ctx.is_first() * (real - first));
  ctx.emit(last * (L(HN_CNT) - one));
  // a read goes on word by word until its count is used up; what follows a read's last word starts another (or is padding)
  const F go = ctx.is_trans() * (real - last);
  ctx.emit(go * (one - ctx.next(HN_IS_REAL)));
  ctx.emit(go * (ctx.next(HN_ADDR) - L(HN_ADDR) - ZKSP_K(4)));
  ctx.emit(go * (ctx.next(HN_CNT) - L(HN_CNT) + one));
  ctx.emit(ctx.is_trans() * (ctx.next(HN_FIRST) - last * ctx.next(HN_IS_REAL)));
  ctx.emit(ctx.is_last() * (real - last));
}
constexpr int kHintConstraints = 13;

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
  // mulh (SH) / mulhsu (SHU): the signed high word from the unsigned one
  const F sh = L(MU_SH), shu = L(MU_SHU), sg = sh + shu;
  ctx.emit(bool_c(sh, one));
  ctx.emit(bool_c(shu, one));
  for (int i = 0; i < 4; ++i) ctx.emit(bool_c(L(MU_K0 + i), one));
  ctx.emit(bool_c(L(MU_HI) + sg, one));  // at most one of mulhu, mulh, mulhsu
  ctx.emit(sg * (one - L(MU_IS_REAL)));
  {
    const F b31 = L(MU_B + 31), c31 = L(MU_C + 31), k0 = L(MU_K0) + L(MU_K0 + 1), k1 = L(MU_K1) + L(MU_K1 + 1);
    const F b_lo = limb_of<F>(ctx, MU_B, 0), b_hi = limb_of<F>(ctx, MU_B, 1), c_lo = limb_of<F>(ctx, MU_C, 0), c_hi = limb_of<F>(ctx, MU_C, 1);
    ctx.emit(sg * (L(MU_R) - limb_of<F>(ctx, MU_P, 2) - k65536 * k0) + sg * b31 * c_lo + sh * c31 * b_lo);
    ctx.emit(sg * (L(MU_R + 1) + k0 - limb_of<F>(ctx, MU_P, 3) - k65536 * k1) + sg * b31 * c_hi + sh * c31 * b_hi);
  }
}
constexpr int kMulConstraints = 166 + 10;

// ---- divider chip: 52 constraints (layout comment above) ----
template <class Ctx>
ZKSP_HD void eval_div(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1), real = L(DV_IS_REAL), k65536 = ZKSP_K(65536);
  ctx.emit(bool_c(real, one));
  F fsum = one - one;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ctx.emit(bool_c(L(DV_F + k), one));
    fsum = fsum + L(DV_F + k);
  }
  ctx.emit(fsum - real);
  {
    constexpr int bools[14] = {DV_SN, DV_SD, DV_SQ, DV_SR, DV_CN, DV_CD, DV_CQ, DV_CR, DV_K, DV_BE, DV_NZD, DV_NZQ, DV_NZR, DV_XS};
#pragma unroll
    for (int k = 0; k < 14; ++k) ctx.emit(bool_c(L(bools[k]), one));
  }
  const F sgn = L(DV_F + 0) + L(DV_F + 2);  // div, rem: signed
  const F sn = L(DV_SN), sd = L(DV_SD), sq = L(DV_SQ), sr = L(DV_SR);
  ctx.emit(sn * (one - sgn));
  ctx.emit(sd * (one - sgn));
  // the sign bits are the operands' top bits (NH, DH: the high limbs without them, looked up below 2^15 for signed operations)
  ctx.emit(L(DV_N + 1) - ZKSP_K(32768) * sn - L(DV_NH));
  ctx.emit(L(DV_D + 1) - ZKSP_K(32768) * sd - L(DV_DH));
  // absolute values: X = AX, or X + AX = 2^32 (limb by limb with a carry) where the sign is set
  {
    constexpr int xs[4] = {DV_N, DV_D, DV_Q, DV_R}, ax[4] = {DV_AN, DV_AD, DV_AQ, DV_AR}, cx[4] = {DV_CN, DV_CD, DV_CQ, DV_CR};
    const F sx[4] = {sn, sd, sq, sr};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const F x_lo = L(xs[k]), x_hi = L(xs[k] + 1), a_lo = L(ax[k]), a_hi = L(ax[k] + 1), c = L(cx[k]);
      ctx.emit(sx[k] * (x_lo + a_lo - k65536 * c) + (one - sx[k]) * (x_lo - a_lo));
      ctx.emit(sx[k] * (x_hi + a_hi + c - k65536) + (one - sx[k]) * (x_hi - a_hi));
    }
  }
  ctx.emit(L(DV_XS) - (sn + sd - (sn * sd).dbl()));
  // zero tests (the limbs are canonical, so their field sum is zero only if both are)
  const F nzd = L(DV_NZD), nzq = L(DV_NZQ), nzr = L(DV_NZR);
  {
    const F dsum = L(DV_D) + L(DV_D + 1), qsum = L(DV_AQ) + L(DV_AQ + 1), rsum = L(DV_AR) + L(DV_AR + 1);
    ctx.emit(dsum * L(DV_INVD) - nzd);
    ctx.emit((one - nzd) * dsum);
    ctx.emit(nzd * (one - real));
    ctx.emit(qsum * L(DV_INVQ) - nzq);
    ctx.emit((one - nzq) * qsum);
    ctx.emit(rsum * L(DV_INVR) - nzr);
    ctx.emit((one - nzr) * rsum);
  }
  // a non-zero divisor: |n| = |q| |d| + |r| (the product's low word from the multiplier chip, its high word zero), |r| < |d|,
  // and the signs
  ctx.emit(nzd * (L(DV_PL) + L(DV_AR) - L(DV_AN) - k65536 * L(DV_K)));
  ctx.emit(nzd * (L(DV_PL + 1) + L(DV_AR + 1) + L(DV_K) - L(DV_AN + 1)));
  ctx.emit(nzd * (L(DV_AD) - L(DV_AR) - one + k65536 * L(DV_BE) - L(DV_E)));
  ctx.emit(nzd * (L(DV_AD + 1) - L(DV_AR + 1) - L(DV_BE) - L(DV_E + 1)));
  ctx.emit(nzd * (sq - L(DV_XS) * nzq));
  ctx.emit(nzd * (sr - sn * nzr));
  // a zero divisor: q = 0xffffffff, r = n
  const F zd = real - nzd;
  ctx.emit(zd * (L(DV_Q) - ZKSP_K(65535)));
  ctx.emit(zd * (L(DV_Q + 1) - ZKSP_K(65535)));
  ctx.emit(zd * (L(DV_R) - L(DV_N)));
  ctx.emit(zd * (L(DV_R + 1) - L(DV_N + 1)));
  // the result: the quotient (div, divu) or the remainder (rem, remu)
  {
    const F wq = L(DV_F + 0) + L(DV_F + 1), wr = L(DV_F + 2) + L(DV_F + 3);
    ctx.emit(L(DV_A) - wq * L(DV_Q) - wr * L(DV_R));
    ctx.emit(L(DV_A + 1) - wq * L(DV_Q + 1) - wr * L(DV_R + 1));
  }
}
constexpr int kDivConstraints = 52;

// ---- ALU chip: 116 constraints in a fixed index space, evaluated as two tasks over disjoint work ----
//   0 is_real, 1..4 selectors, 5..36 B bits, 37..68 C bits, 69..100 X bits, 101..102 K0 K1, 103 one selector per real row,
//   104..111 shifts, 112..115 signed less-than
//   task 0  selectors, C bit by bit: booleans
//   task 1  X with B, C: shifts, less-than
namespace aluidx {
constexpr int kSel = 1, kBoolB = 5, kBoolC = 37, kBoolX = 69, kBoolK = 101, kSelSum = 103, kShift = 104, kCmp = 112;
}
constexpr int kAluTasks = 2;
template <int TASK, class Ctx>
ZKSP_HD void eval_alu_task(Ctx& ctx) {
  using F = typename Ctx::F;
  using namespace aluidx;
  const F one = ctx.k(kR1), zero = one - one;
  const F k65536 = ZKSP_K(65536);
#define OPF(o) ctx.local(AL_SEL + (o) - SLL)
  const F a_lo = L(AL_A), a_hi = L(AL_A + 1);
  if (TASK == 0) {
    const F is_real = L(AL_IS_REAL);
    ctx.emit_at(0, bool_c(is_real, one));
    F selsum = zero;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const F v = L(AL_SEL + k);
      ctx.emit_at(kSel + k, bool_c(v, one));
      selsum = selsum + v;
    }
    ctx.emit_at(kSelSum, selsum - is_real);
    ctx.emit_at(kBoolK, bool_c(L(AL_K0), one));
    ctx.emit_at(kBoolK + 1, bool_c(L(AL_K1), one));
#pragma unroll
    for (int i = 0; i < 32; ++i) ctx.emit_at(kBoolC + i, bool_c(L(AL_C + i), one));
  }
  if (TASK == 1) {
    F samt = L(AL_C + 4);
#pragma unroll
    for (int i = 3; i >= 0; --i) samt = samt.dbl() + L(AL_C + i);
    F c_lo = L(AL_C + 15), c_hi = L(AL_C + 31);
    const F c31 = c_hi;
#pragma unroll
    for (int i = 14; i >= 0; --i) {
      c_lo = c_lo.dbl() + L(AL_C + i);
      c_hi = c_hi.dbl() + L(AL_C + 16 + i);
    }
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
        const F bi = L(AL_B + i);
        ctx.emit_at(kBoolB + i, bool_c(bi, one));
        ctx.stash(i, bi);
        p32 = p32 + pw * bi;
        if (i == 15) p16 = p32;
        if (i == 31) b31 = bi;
        pw = pw.dbl();
      }
    }
    const F inv2 = ctx.k(cmonty(inv_pow2_mod(1))), inv2_16 = ctx.k(cmonty(inv_pow2_mod(16)));
    const F b_lo = p16, b_hi = (p32 - p16) * inv2_16;
    F sum = zero, idx = zero, x_lo = zero, x_hi = zero;
    F sll_lo = zero, sll_hi = zero, srl_lo = zero, srl_hi = zero, fill_lo = zero, fill_hi = zero;
    {
      F pa = zero, pb = p16, pc = p16, pd = p32;        // P_k, P_min(16+k,32), P_max(16-k,0), P_(32-k)
      F pw = one, ipw = one, kf = zero;                  // 2^k, 2^-k, k
      F d15 = ctx.k(cmonty(pow2_mod(15))), d31 = ctx.k(cmonty(pow2_mod(31)));  // 2^(15-k), 2^(31-k)
      for (int k = 0; k < 32; ++k) {
        const F xk = L(AL_X + k);
        ctx.emit_at(kBoolX + k, bool_c(xk, one));
        sum = sum + xk;
        idx = idx + kf * xk;
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
    // signed less-than: X = B - C (mod 2^32) with the sign bits swapped, K1 = "less than"
    const F k0 = L(AL_K0), k1 = L(AL_K1), cmp = OPF(SLT);
    ctx.emit_at(kCmp + 0, cmp * (b_lo - c_lo + k65536 * k0 - x_lo));
    ctx.emit_at(kCmp + 1, cmp * (b_hi - c_hi - k0 + k65536 * k1 - x_hi + k65536 * (c31 - b31)));
    ctx.emit_at(kCmp + 2, cmp * (a_lo - k1));
    ctx.emit_at(kCmp + 3, cmp * a_hi);
  }
#undef OPF
}
constexpr int kAluConstraints = 116;
template <class Ctx>
ZKSP_HD void eval_alu(Ctx& ctx) {
  eval_alu_task<0>(ctx);
  eval_alu_task<1>(ctx);
  ctx.set_count(kAluConstraints);
}

// ---- bitwise chip: one operation per real row; the byte lookups do the rest ----
template <class Ctx>
ZKSP_HD void eval_bw(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1), is_real = L(BW_IS_REAL);
  ctx.emit(bool_c(is_real, one));
  F selsum = one - one;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const F v = L(BW_SEL + k);
    ctx.emit(bool_c(v, one));
    selsum = selsum + v;
  }
  ctx.emit(selsum - is_real);
}
constexpr int kBwConstraints = 5;

// ---- sub-word chip: M is the memory word, C the low limb of the stored register, both as bytes (looked up in the table
// chip's byte-pair rows); S the sign bit a signed load extends, bit 7 of the byte SELB (byte-operation lookup) ----
template <class Ctx>
ZKSP_HD void eval_sub(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1), zero = one - one;
  enum { kLB = 0, kLH, kLBU, kLHU, kSB, kSH };
#define SF(k) ctx.local(SW_SEL + (k))
  const F is_real = L(SW_IS_REAL);
  ctx.emit(bool_c(is_real, one));
  F selsum = zero, osum = zero;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const F v = SF(k);
    ctx.emit(bool_c(v, one));
    selsum = selsum + v;
  }
  F o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[k] = L(SW_O + k);
    ctx.emit(bool_c(o[k], one));
    osum = osum + o[k];
  }
  const F sgn = L(SW_S), selb = L(SW_SELB);
  ctx.emit(bool_c(sgn, one));
  ctx.emit(selsum - is_real);
  ctx.emit(osum - is_real);
  const F mb[4] = {L(SW_MB), L(SW_MB + 1), L(SW_MB + 2), L(SW_MB + 3)}, cb = L(SW_CB);
  const F k256 = ZKSP_K(256), k65535 = ZKSP_K(65535);
  const F m_lo = mb[0] + k256 * mb[1], m_hi = mb[2] + k256 * mb[3], c_lo = cb + k256 * L(SW_CB + 1);
  const F a_lo = L(SW_A), a_hi = L(SW_A + 1);
  // half-word accesses are 2-aligned
  ctx.emit((SF(kLH) + SF(kLHU) + SF(kSH)) * (o[1] + o[3]));
  // the sign: only signed loads have one; it belongs to the accessed byte / the accessed half-word's upper byte
  F bv = zero;
#pragma unroll
  for (int p = 0; p < 4; ++p) bv = bv + o[p] * mb[p];
  const F hv = o[0] * m_lo + o[2] * m_hi, hb = o[0] * mb[1] + o[2] * mb[3];
  ctx.emit((one - SF(kLB) - SF(kLH)) * sgn);
  ctx.emit(SF(kLB) * (selb - bv));
  ctx.emit(SF(kLH) * (selb - hb));
  ctx.emit(SF(kLHU) * (a_lo - hv));
  ctx.emit(SF(kLHU) * a_hi);
  ctx.emit(SF(kLH) * (a_lo - hv));
  ctx.emit(SF(kLH) * (a_hi - k65535 * sgn));
  ctx.emit(SF(kLBU) * (a_lo - bv));
  ctx.emit(SF(kLBU) * a_hi);
  ctx.emit(SF(kLB) * (a_lo - (bv + ZKSP_K(0xff00) * sgn)));
  ctx.emit(SF(kLB) * (a_hi - k65535 * sgn));
  // stores: A is the word left behind - the old word with the low half-word / byte of the stored register put in
  ctx.emit(SF(kSH) * (a_lo - m_lo - o[0] * (c_lo - m_lo)));
  ctx.emit(SF(kSH) * (a_hi - m_hi - o[2] * (c_lo - m_hi)));
  ctx.emit(SF(kSB) * (a_lo - m_lo - (o[0] * (cb - mb[0]) + k256 * (o[1] * (cb - mb[1])))));
  ctx.emit(SF(kSB) * (a_hi - m_hi - (o[2] * (cb - mb[2]) + k256 * (o[3] * (cb - mb[3])))));
  // loads put no register limb on the bus
  ctx.emit((SF(kLB) + SF(kLH) + SF(kLBU) + SF(kLHU)) * c_lo);
#undef SF
}
constexpr int kSubConstraints = 31;

// ---- Poseidon2 chip: every S-box through its cube (x^3, then x^7 = (x^3)^2 x: degree 3); between S-boxes the state is
// linear in the columns.  Ctx::p2(): the permutation's constants (Montgomery words).  286 permutation constraints, then the
// 67 that tie the rows of an opening together (layout comment above). ----
template <class F>
ZKSP_HD void p2air_external_linear(F* s) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const F a = s[4 * c], b = s[4 * c + 1], cc = s[4 * c + 2], d = s[4 * c + 3];
    const F sum = (a + b) + (cc + d);
    s[4 * c] = sum + a + b.dbl();       // 2a + 3b + c + d
    s[4 * c + 1] = sum + b + cc.dbl();  // a + 2b + 3c + d
    s[4 * c + 2] = sum + cc + d.dbl();  // a + b + 2c + 3d
    s[4 * c + 3] = sum + d + a.dbl();   // 3a + b + c + 2d
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const F col = (s[j] + s[4 + j]) + (s[8 + j] + s[12 + j]);
#pragma unroll
    for (int c = 0; c < 4; ++c) s[4 * c + j] = s[4 * c + j] + col;
  }
}
// extension-field values over the constraint field F (F_p[x] / (x^4 - 11)): the device instantiates F = Fp (a lane is an LDE
// point), the verifier F = Fp4 (the point zeta)
template <class F>
struct X4 {
  F c[4];
};
template <class F>
ZKSP_HD X4<F> x4_add(const X4<F>& a, const X4<F>& b) { X4<F> r; for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] + b.c[i]; return r; }
template <class F>
ZKSP_HD X4<F> x4_sub(const X4<F>& a, const X4<F>& b) { X4<F> r; for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] - b.c[i]; return r; }
template <class F>
ZKSP_HD X4<F> x4_scale(const X4<F>& a, const F& b) { X4<F> r; for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] * b; return r; }
template <class F>
ZKSP_HD X4<F> x4_mul(const X4<F>& a, const X4<F>& b, const F& k11) {
  X4<F> r;
  r.c[0] = a.c[0] * b.c[0] + k11 * (a.c[1] * b.c[3] + a.c[2] * b.c[2] + a.c[3] * b.c[1]);
  r.c[1] = a.c[0] * b.c[1] + a.c[1] * b.c[0] + k11 * (a.c[2] * b.c[3] + a.c[3] * b.c[2]);
  r.c[2] = a.c[0] * b.c[2] + a.c[1] * b.c[1] + a.c[2] * b.c[0] + k11 * (a.c[3] * b.c[3]);
  r.c[3] = a.c[0] * b.c[3] + a.c[1] * b.c[2] + a.c[2] * b.c[1] + a.c[3] * b.c[0];
  return r;
}
template <class F, class Ctx>
ZKSP_HD X4<F> x4_local(const Ctx& ctx, int col) { X4<F> r; for (int i = 0; i < 4; ++i) r.c[i] = ctx.local(col + i); return r; }
template <class F, class Ctx>
ZKSP_HD X4<F> x4_next(const Ctx& ctx, int col) { X4<F> r; for (int i = 0; i < 4; ++i) r.c[i] = ctx.next(col + i); return r; }

// The permutation of one row: the columns IN (16 words), EXT (8 rounds x (16 cubes, 16 seventh powers)), INT (13 x (cube,
// seventh power)); 282 constraints; st[] comes back as the 16 output words, linear in the last round's columns.
template <class Ctx>
ZKSP_HD void p2_perm_constraints(Ctx& ctx, int c_in, int c_ext, int c_int, typename Ctx::F* st) {
  using F = typename Ctx::F;
  const P2Consts* kc = ctx.p2();
#pragma unroll
  for (int i = 0; i < 16; ++i) st[i] = ctx.local(c_in + i);
  p2air_external_linear(st);
  for (int rd = 0; rd < 8; ++rd) {
    if (rd == 4) {  // the 13 internal rounds sit between the two halves of the external ones
      for (int ir = 0; ir < 13; ++ir) {
        const F x = st[0] + ctx.k(kc->internal[ir]), x3 = ctx.local(c_int + 2 * ir), y = ctx.local(c_int + 2 * ir + 1);
        ctx.emit(x3 - x * x * x);
        ctx.emit(y - x3 * x3 * x);
        st[0] = y;
        F sum = st[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) sum = sum + st[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = st[i] * ctx.k(kc->diag[i]) + sum;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const F x = st[i] + ctx.k(kc->ext[rd][i]), x3 = ctx.local(c_ext + 32 * rd + i), y = ctx.local(c_ext + 32 * rd + 16 + i);
      ctx.emit(x3 - x * x * x);
      ctx.emit(y - x3 * x3 * x);
      st[i] = y;
    }
    p2air_external_linear(st);
  }
}
template <class Ctx>
ZKSP_HD void eval_p2(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1), real = L(P2_IS_REAL);
  ctx.emit(bool_c(real, one));
  ctx.emit(ctx.is_trans() * ctx.next(P2_IS_REAL) * (one - real));  // the real rows are a prefix
  F st[16];
  p2_perm_constraints(ctx, P2_IN, P2_EXT, P2_INT, st);
  // st[] now holds the permutation's 16 output words, linear in the last round's columns.
  // Row kinds: one per real row; NEW lives on sponge rows, FR on first blocks, SND (the hash goes to an injection row) and SE
  // (the Horner sum goes to the query chip) on sponge rows, RE (the end of a run) on path and injection rows.
  const F fn = L(P2_FN), sz = L(P2_SZ), sc = L(P2_SC), pl = L(P2_PL), pr = L(P2_PR), fj = L(P2_FJ), nw = L(P2_NEW), snd = L(P2_SND),
          fr = L(P2_FR);
  ctx.emit(bool_c(fn, one)); ctx.emit(bool_c(sz, one)); ctx.emit(bool_c(sc, one)); ctx.emit(bool_c(pl, one)); ctx.emit(bool_c(pr, one));
  ctx.emit(bool_c(fj, one)); ctx.emit(bool_c(nw, one)); ctx.emit(bool_c(snd, one)); ctx.emit(bool_c(fr, one));
  const F chain = sc + pl + pr + fj;  // the kinds that take over from the row before
  ctx.emit(fn + sz + chain - real);
  ctx.emit(nw * (one - sz - sc));
  ctx.emit(snd * (one - sz - sc));
  ctx.emit(fr * (one - sz));
  // a first block starts from the zero state; the first block of a run from K = 1, M = 0
#pragma unroll
  for (int i = 0; i < 8; ++i) ctx.emit(sz * L(P2_IN + 8 + i));
  const F key = L(P2_KL) + ZKSP_K(65536) * L(P2_KH), m = L(P2_M);
  ctx.emit(sz * nw * (key - one));
  ctx.emit(sz * nw * m);
  ctx.emit(ctx.is_first() * chain);  // nothing precedes row 0 (so the constraints below hold cyclically)
  // the next row, where it takes over from this one
  const F nsc = ctx.next(P2_SC), npl = ctx.next(P2_PL), npr = ctx.next(P2_PR), nfj = ctx.next(P2_FJ), npath = npl + npr;
  const F nkey = ctx.next(P2_KL) + ZKSP_K(65536) * ctx.next(P2_KH), nm = ctx.next(P2_M);
  ctx.emit((nsc + npath + nfj) * (ctx.next(P2_T) - L(P2_T)));
  // ... a sponge goes on: the capacity, the labels and the flag of its run; only a sponge row precedes it
#pragma unroll
  for (int i = 0; i < 8; ++i) ctx.emit(nsc * (ctx.next(P2_IN + 8 + i) - st[8 + i]));
  ctx.emit(nsc * (nkey - key));
  ctx.emit(nsc * (nm - m));
  ctx.emit(nsc * (ctx.next(P2_NEW) - nw));
  ctx.emit(nsc * (one - sz - sc));
  // ... a path step: the running digest on its side, one more position bit, one more level; it follows the leaf's sponge
  // (a run's, not an injected matrix's), a path step or an injection
#pragma unroll
  for (int i = 0; i < 8; ++i) ctx.emit(npl * (ctx.next(P2_IN + i) - st[i]));
#pragma unroll
  for (int i = 0; i < 8; ++i) ctx.emit(npr * (ctx.next(P2_IN + 8 + i) - st[i]));
  ctx.emit(npath * (nkey - key.dbl() - npr));
  ctx.emit(npath * (nm - m.dbl()));
  ctx.emit(npath * (one - (sz + sc) * nw - pl - pr - fj));
  // ... an injection: the running digest on the left, the level marked; only a path step precedes it
#pragma unroll
  for (int i = 0; i < 8; ++i) ctx.emit(nfj * (ctx.next(P2_IN + i) - st[i]));
  ctx.emit(nfj * (nkey - key));
  ctx.emit(nfj * (nm - m - one));
  ctx.emit(nfj * (one - pl - pr));
  // ---- format v16 (stage 2b) ----
  // the end of a run is a path step or an injection; the end of a matrix row's hash is a sponge row that no block follows
  const F re = L(P2_RE), se = L(P2_SE);
  ctx.emit(bool_c(re, one));
  ctx.emit(bool_c(se, one));
  ctx.emit(re * (one - pl - pr - fj));
  ctx.emit(se * (one - sz - sc));
  ctx.emit(se * nsc);
  // Horner's rule in alpha_f over the absorbed words, block by block: SO' = SO alpha^8 + sum_{i < 8} alpha^(7 - i) in_i.
  // AP holds alpha^1 .. alpha^8 (each the one before times AP[0]); a sponge's rows share them.
  const F k11 = ZKSP_K(11);
  X4<F> ap[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ap[j] = x4_local<F>(ctx, P2_AP + 4 * j);
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const X4<F> pj = x4_mul(ap[j], ap[0], k11);
#pragma unroll
    for (int i = 0; i < 4; ++i) ctx.emit(ap[j + 1].c[i] - pj.c[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) ctx.emit(nsc * (ctx.next(P2_AP + i) - ap[0].c[i]));
  {
    X4<F> bv;  // this row's block
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      F v = ap[6].c[c] * L(P2_IN);
#pragma unroll
      for (int i = 1; i < 7; ++i) v = v + ap[6 - i].c[c] * L(P2_IN + i);
      bv.c[c] = v;
    }
    bv.c[0] = bv.c[0] + L(P2_IN + 7);
#pragma unroll
    for (int c = 0; c < 4; ++c) ctx.emit(sz * (L(P2_SO + c) - bv.c[c]));
    // the next row's block over the next row's powers (equal to this row's where the sponge goes on)
    X4<F> nap7 = x4_next<F>(ctx, P2_AP + 28), so = x4_local<F>(ctx, P2_SO);
    const X4<F> carried = x4_mul(so, nap7, k11);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      F v = ctx.next(P2_AP + 24 + c) * ctx.next(P2_IN);
#pragma unroll
      for (int i = 1; i < 7; ++i) v = v + ctx.next(P2_AP + 4 * (6 - i) + c) * ctx.next(P2_IN + i);
      if (c == 0) v = v + ctx.next(P2_IN + 7);
      ctx.emit(nsc * (ctx.next(P2_SO + c) - carried.c[c] - v));
    }
  }
}
constexpr int kP2Constraints = 2 + 8 * 32 + 13 * 2 + 67 + 5 + 28 + 4 + 4 + 4;

// ---- query chip: the constraints in the order of the layout comment above (DESIGN.md "Query chip") ----
template <class Ctx>
ZKSP_HD void eval_qr(Ctx& ctx) {
  using F = typename Ctx::F;
#define N(c) ctx.next(c)
  const F one = ctx.k(kR1), real = L(QR_IS_REAL), first = L(QR_FIRST), last = L(QR_LAST), bit = L(QR_BIT), eq = L(QR_EQ),
          f1 = L(QR_F1), f2 = L(QR_F2), f3 = L(QR_F3), csr = L(QR_CSR), fl = L(QR_FL), lay = L(QR_LAY), cs = L(QR_CS), pr0 = L(QR_PR0),
          p0a = L(QR_P0A), hasro = L(QR_HASRO), has0 = L(QR_HAS0);
  {
    const F bools[17] = {real, first, last, bit, eq, f1, f2, f3, csr, fl, lay, cs, pr0, p0a, hasro, has0, L(QR_CNT0)};
#pragma unroll
    for (int i = 0; i < 17; ++i) ctx.emit(bool_c(bools[i], one));
  }
  const F zero = one - one;
  const F cont = N(QR_IS_REAL) - N(QR_FIRST);  // the next row goes on with this query
  const F nl1 = N(QR_LAY) - N(QR_FL);         // ... and is a layer row other than layer 0 (so this row is a layer row too)
  ctx.emit(ctx.is_trans() * N(QR_IS_REAL) * (one - real));  // the real rows are a prefix
  ctx.emit(ctx.is_first() * (real - first));
  ctx.emit((first + last + csr + fl + lay + pr0 + p0a + hasro + has0 + bit + eq + f1 + f2 + f3 + cs) * (one - real));
  // 31 rows per query: bit 30 down to bit 0
  const F j = L(QR_J);
  ctx.emit(first * (j - ZKSP_K(30)));
  ctx.emit(last * j);
  ctx.emit(cont * (N(QR_J) - j + one));
  ctx.emit(last * cont);
  ctx.emit((real - last) * (one - cont));
  ctx.emit(cont * (N(QR_LEAF) - L(QR_LEAF)));
  ctx.emit(cont * (N(QR_QL) - L(QR_QL)));
  // the word from its most significant bit; canonical: it does not exceed p - 1 = 0x78000000 (EQ: the bits so far are p - 1's)
  ctx.emit(first * (L(QR_ACC) - bit));
  ctx.emit(cont * (N(QR_ACC) - L(QR_ACC).dbl() - N(QR_BIT)));
  ctx.emit(first * (f1 + f2 + f3));
  ctx.emit(cont * (N(QR_F1) - first));
  ctx.emit(cont * (N(QR_F2) - f1));
  ctx.emit(cont * (N(QR_F3) - f2));
  ctx.emit(first * (eq - bit));
  {
    const F nf = N(QR_F1) + N(QR_F2) + N(QR_F3);
    ctx.emit(nf * (N(QR_EQ) - eq * N(QR_BIT)));
    ctx.emit((cont - nf) * (N(QR_EQ) - eq));
    ctx.emit((real - first - f1 - f2 - f3) * eq * bit);
  }
  // row kinds: the rows above the coset bit, the coset bit's row (CSR), the layer rows (FL: layer 0); the last row is a layer row
  ctx.emit(csr * lay);
  ctx.emit(fl * (one - lay));
  ctx.emit(first * (csr + lay));
  ctx.emit(cont * (N(QR_FL) - csr));
  ctx.emit(cont * (N(QR_LAY) - csr - lay));
  ctx.emit(last * (one - lay));
  ctx.emit(fl * L(QR_K));
  ctx.emit(nl1 * (N(QR_K) - L(QR_K) - one));
  ctx.emit(csr * (cs - bit));
  ctx.emit(cont * (csr + lay) * (N(QR_CS) - cs));
  // 2^J, the bits below J as a number (LOW) and in reverse (REV): the positions of the openings are linear in them
  ctx.emit(last * (L(QR_POW) - one));
  ctx.emit(last * L(QR_LOW));
  ctx.emit(last * L(QR_REV));
  ctx.emit(cont * (L(QR_POW) - N(QR_POW).dbl()));
  ctx.emit(cont * (L(QR_LOW) - N(QR_LOW) - N(QR_BIT) * N(QR_POW)));
  ctx.emit(cont * (L(QR_REV) - N(QR_REV).dbl() - N(QR_BIT)));
  // the preprocessed tree is 2^16 tall: its opening is received on the row of bit 16, exactly once
  ctx.emit(pr0 * (j - ZKSP_K(16)));
  ctx.emit(first * (L(QR_CNT0) - pr0));
  ctx.emit(cont * (N(QR_CNT0) - L(QR_CNT0) - N(QR_PR0)));
  ctx.emit(last * (L(QR_CNT0) - one));
  ctx.emit(first * p0a);
  ctx.emit(cont * (N(QR_P0A) - p0a - pr0));
  ctx.emit(cont * pr0 * (N(QR_KEY0) - one));
  ctx.emit(cont * pr0 * N(QR_M0));
  ctx.emit(cont * p0a * (N(QR_KEY0) - L(QR_KEY0).dbl() - bit));
  ctx.emit(cont * p0a * (N(QR_M0) - L(QR_M0).dbl() - N(QR_HAS0)));
  ctx.emit(has0 * (one - p0a));
  ctx.emit(has0 * (one - hasro));
  ctx.emit(cont * (N(QR_MT0) - L(QR_MT0)));
  ctx.emit(last * (L(QR_MT0) - ZKSP_K(4) * L(QR_M0)));
  // keys and masks of the injected rows of the other three trees (as tall as the proof): the position bits above, and which
  // heights joined on the way
  ctx.emit(fl * (L(QR_KEYJ) - one));
  ctx.emit(fl * L(QR_MJ));
  ctx.emit(nl1 * (N(QR_KEYJ) - L(QR_KEYJ).dbl() - bit));
  ctx.emit(nl1 * (N(QR_MJ) - L(QR_MJ).dbl() - N(QR_HASRO)));
  ctx.emit(cont * (N(QR_MT) - L(QR_MT)));
  ctx.emit(last * (L(QR_MT) - ZKSP_K(4) * L(QR_MJ)));
  ctx.emit(hasro * (one - lay));
  ctx.emit(fl * (one - hasro));
  // the domain points: omega^-1 (the verifier's, of order 2^(lm + 1)) to the power cs + 2 m by square and multiply (R: omega^-m
  // so far), its squares down the layers (YKI), the shift g^-(2^k) (GI), the inverse of the layer's point (XINV)
  const F omi = L(QR_OMI);
  ctx.emit(cont * (N(QR_OMI) - omi));
  ctx.emit(L(QR_MU) - real - bit * (omi - one));  // (1 on a real row whose bit is clear; 0 on padding rows)
  ctx.emit(L(QR_CSM) - real - cs * (omi - one));
  ctx.emit(L(QR_R2) - L(QR_R) * L(QR_R));
  ctx.emit(fl * (L(QR_R) - L(QR_MU)));
  ctx.emit(nl1 * (N(QR_R) - L(QR_R2) * N(QR_MU)));
  ctx.emit(cont * (N(QR_YT) - L(QR_YT)));
  ctx.emit(last * (L(QR_YT) - L(QR_R2) * L(QR_CSM)));
  ctx.emit(fl * (L(QR_YKI) - L(QR_YT)));
  ctx.emit(nl1 * (N(QR_YKI) - L(QR_YKI) * L(QR_YKI)));
  ctx.emit(fl * (L(QR_GI) - ctx.k(cmonty(kGenInv))));
  ctx.emit(nl1 * (N(QR_GI) - L(QR_GI) * L(QR_GI)));
  ctx.emit(L(QR_XINV) - L(QR_GI) * L(QR_YKI) * (one - bit.dbl()));
  // the fold
  const F k11 = ZKSP_K(11);
  const X4<F> lo = x4_local<F>(ctx, QR_LO), hi = x4_local<F>(ctx, QR_HI), be = x4_local<F>(ctx, QR_BETA), d = x4_sub(lo, hi);
#pragma unroll
  for (int i = 0; i < 4; ++i) ctx.emit(L(QR_E + i) - lo.c[i] + bit * d.c[i]);
  {
    const X4<F> pd = x4_mul(be, d, k11);
    const F xinv = L(QR_XINV);
#pragma unroll
    for (int i = 0; i < 4; ++i) ctx.emit(L(QR_F + i).dbl() - lo.c[i] - hi.c[i] - xinv * pd.c[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) ctx.emit((one - hasro) * L(QR_RO + i));
#pragma unroll
  for (int i = 0; i < 4; ++i) ctx.emit(fl * (L(QR_E + i) - L(QR_RO + i)));
#pragma unroll
  for (int i = 0; i < 4; ++i) ctx.emit(nl1 * (N(QR_E + i) - L(QR_F + i) - N(QR_RO + i)));
  // the reduced opening of the height that joins on this row (DESIGN.md "Reduced openings"): the Horner sums H_r of the
  // four trees' opened rows, delta between the trees, the verifier's constants B1, B2, w_H
  {
    const X4<F> dl = x4_local<F>(ctx, QR_DL), d2 = x4_local<F>(ctx, QR_D2), d3 = x4_local<F>(ctx, QR_D3), d4 = x4_local<F>(ctx, QR_D4),
                g2 = x4_local<F>(ctx, QR_G2), zeta = x4_local<F>(ctx, QR_ZETA), zw = x4_local<F>(ctx, QR_ZW), d0 = x4_local<F>(ctx, QR_D0),
                d1 = x4_local<F>(ctx, QR_D1), h0 = x4_local<F>(ctx, QR_H), h1 = x4_local<F>(ctx, QR_H + 4), h2 = x4_local<F>(ctx, QR_H + 8),
                h3 = x4_local<F>(ctx, QR_H + 12);
    const X4<F> e2 = x4_mul(dl, dl, k11), e3 = x4_mul(d2, dl, k11), e4 = x4_mul(d2, d2, k11), eg = x4_add(h1, x4_mul(dl, h2, k11));
    const X4<F> ez = x4_scale(zeta, L(QR_WH));
#pragma unroll
    for (int i = 0; i < 4; ++i) ctx.emit(d2.c[i] - e2.c[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) ctx.emit(d3.c[i] - e3.c[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) ctx.emit(d4.c[i] - e4.c[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) ctx.emit(g2.c[i] - eg.c[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) ctx.emit(zw.c[i] - ez.c[i]);
    // D0 (g y - zeta) = 1 and D1 (g y - zeta w) = 1, multiplied through by 1 / y (the column YKI)
    const F g = ctx.k(cmonty(kGen)), yki = L(QR_YKI);
    X4<F> den0 = x4_scale(zeta, zero - yki), den1 = x4_scale(zw, zero - yki);
    den0.c[0] = den0.c[0] + g;
    den1.c[0] = den1.c[0] + g;
    const X4<F> p0 = x4_mul(d0, den0, k11), p1 = x4_mul(d1, den1, k11);
#pragma unroll
    for (int i = 0; i < 4; ++i) ctx.emit(p0.c[i] - (i == 0 ? yki : zero));
#pragma unroll
    for (int i = 0; i < 4; ++i) ctx.emit(p1.c[i] - (i == 0 ? yki : zero));
    X4<F> hs = x4_add(x4_add(h0, x4_mul(dl, h1, k11)), x4_add(x4_mul(d2, h2, k11), x4_mul(d3, h3, k11)));
    hs = x4_sub(hs, x4_local<F>(ctx, QR_B1));
    const X4<F> t2 = x4_sub(x4_mul(d4, g2, k11), x4_local<F>(ctx, QR_B2));
    const X4<F> ro = x4_add(x4_mul(d0, hs, k11), x4_mul(d1, t2, k11));
#pragma unroll
    for (int i = 0; i < 4; ++i) ctx.emit(L(QR_RO + i) - ro.c[i]);
  }
#undef N
}
constexpr int kQrConstraints = 140;  // (counted by running them: the verifier checks the count)

// ---- transcript chip ----
template <class Ctx>
ZKSP_HD void eval_tr(Ctx& ctx) {
  using F = typename Ctx::F;
  const F one = ctx.k(kR1), real = L(TR_IS_REAL), first = L(TR_FIRST), abs_ = L(TR_ABS);
  ctx.emit(bool_c(real, one));
  ctx.emit(ctx.is_trans() * ctx.next(TR_IS_REAL) * (one - real));  // the real rows are a prefix
  F st[16];
  p2_perm_constraints(ctx, TR_IN, TR_EXT, TR_INT, st);
  F fsum = first + abs_;
  ctx.emit(bool_c(first, one));
  ctx.emit(bool_c(abs_, one));
#pragma unroll
  for (int k = 0; k < 7 + 8; ++k) {
    const F v = L(TR_UROOT + k);
    ctx.emit(bool_c(v, one));
    fsum = fsum + v;
  }
  ctx.emit(fsum * (one - real));
  // a transcript starts by absorbing over the zero state, at step 0
  ctx.emit(first * (one - abs_));
  ctx.emit(first * L(TR_STEP));
#pragma unroll
  for (int i = 0; i < 8; ++i) ctx.emit(first * L(TR_IN + 8 + i));
  ctx.emit(ctx.is_first() * (real - first));
  // the next duplex of the same transcript: the capacity goes on; a squeeze takes the whole state over
  const F cont = ctx.next(TR_IS_REAL) - ctx.next(TR_FIRST);
  ctx.emit(cont * (ctx.next(TR_LEAF) - L(TR_LEAF)));
  ctx.emit(cont * (ctx.next(TR_STEP) - L(TR_STEP) - one));
#pragma unroll
  for (int i = 0; i < 8; ++i) ctx.emit(cont * (ctx.next(TR_IN + 8 + i) - st[8 + i]));
  {
    const F sq = cont * (one - ctx.next(TR_ABS));
#pragma unroll
    for (int i = 0; i < 8; ++i) ctx.emit(sq * (ctx.next(TR_IN + i) - st[i]));
  }
  // uses: query words only on query rows; a multiplicity only where the use is flagged
  {
    F qm = L(TR_QM);
#pragma unroll
    for (int k = 1; k < 8; ++k) qm = qm + L(TR_QM + k);
    ctx.emit(qm * (one - L(TR_UQ)));
  }
  ctx.emit(L(TR_MROOT) * (one - L(TR_UROOT)));
  ctx.emit(L(TR_MFIN) * (one - L(TR_UFIN)));
  ctx.emit(L(TR_MZETA) * (one - L(TR_UZETA)));
  ctx.emit(L(TR_MAF) * (one - L(TR_UAF)));
  ctx.emit(L(TR_MBETA) * (one - L(TR_UBETA)));
}
constexpr int kTrConstraints = 2 + 282 + 2 + 15 + 1 + 2 + 8 + 1 + 2 + 8 + 8 + 1 + 5;

// every image word is sent exactly once
template <class Ctx>
ZKSP_HD void eval_image(Ctx& ctx) {
  ctx.emit(L(0) - ctx.prep(IMG_P_REAL));
}
// only multiples of 4 answer aligned lookups, only values 1 .. kAddrHiMax high-address-limb lookups
template <class Ctx>
ZKSP_HD void eval_table(Ctx& ctx) {
  ctx.emit(L(TB_M_AL) * ctx.prep(TB_P_NA));
  ctx.emit(L(TB_M_TOP) * ctx.prep(TB_P_NT));
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
       : chip == kMemFinal ? kMemFinalConstraints : chip == kImage ? 1 : chip == kProgram ? 0 : chip == kMul ? kMulConstraints
       : chip == kTable ? 2 : is_alu_chip(chip) ? kAluConstraints : is_sub_chip(chip) ? kSubConstraints : is_bw_chip(chip) ? kBwConstraints : chip == kP2 ? kP2Constraints : chip == kEcall ? kEcallConstraints : chip == kQr ? kQrConstraints : chip == kTr ? kTrConstraints : chip == kHint ? kHintConstraints : chip == kDiv ? kDivConstraints : 0;
}

}  // namespace mach
}  // namespace zksp
