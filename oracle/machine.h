/* ORACLE -- TEST INFRASTRUCTURE ONLY (see field.h).
 *
 * The multi-chip machine proof: chip layouts, LogUp bus
 * interactions, trace generation and constraint evaluation of this repository's own
 * arithmetisation of RV32IM + the keccak precompile.  It plays the part sp1-core-machine 3.4.0
 * (reference Cargo.lock:7130: CPU, Program, memory, ALU, keccak-permute chips joined by a LogUp
 * lookup argument) plays beneath the reference's `client.prove(&pk, stdin).run()`
 * (prover/src/bin/main.rs:71-74).  PARITY UNPINNED vs SP1: the sources are absent; chip
 * decomposition, column layouts and constraint order are this repository's own (DESIGN.md
 * "Machine proof").  The statement is the reference's: the committed guest
 * (circuits/sp1-merkle-proof/src/main.rs:4-14, crypto-ops/src/lib.rs:8-23) ran to HALT(0) and
 * committed these public values.
 *
 * Format v14 (v12: round 3; v14: round 4 - address floor, leaf-proof check).  The CPU row no longer carries its operands as bits: it holds 16-bit limbs,
 * adds / subtracts / compares (equality, unsigned order) / moves words itself, and sends xor / or /
 * and to a bitwise chip (bytes, looked up in a byte-operation table), shifts and signed less-than to
 * an ALU chip (bits) and every sub-word load or store to a sub-word chip, each with one row per such
 * instruction (SP1's split of the CPU chip from its event-sized ALU chips).  Range discipline: every tuple on the memory bus carries
 * canonical 16-bit limbs because every producer guarantees it (image: preprocessed; free
 * initial values, sums, differences and the hinted length: looked up in a 2^16-row table; ALU,
 * sub-word, multiplier and keccak results: bits), so readers need no decomposition.  Memory
 * addresses and jump targets are kept below 0x78000000 < p by the same table, so the map from
 * 32-bit values to field elements is injective wherever a bus compares them.
 */
#ifndef ZKSP_ORACLE_MACHINE_H
#define ZKSP_ORACLE_MACHINE_H
#include <stddef.h>
#include <stdint.h>

#include "zksp_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- chips, in proof order ---- */
/* ALU, sub-word and bitwise rows are each split over two instances of one AIR: the first has the largest power of two
 * of rows strictly below the count (at least 32), the second the rest rounded up to a power of two.
 * The CPU rows (format v13) are spread over CPU_INST instances of one height, 2^ceil(log2(cycles / CPU_INST)): as many
 * as the cycles need, the others at the minimum height (all padding).  391 400 cycles take 6 x 2^16 rows: the same
 * cells as 2^18 + 2^17, but in rows six times as wide - a Merkle tree, a quotient and a FRI domain of a quarter of
 * the height, and the per-row costs of the commitments (node compressions, the quotient's leaf) shared by six. */
#define CPU_INST 8
enum {
  CH_CPU = 0, CH_KECCAK, CH_KMEM, CH_MEMFINAL, CH_IMAGE, CH_PROGRAM, CH_MUL, CH_TABLE, CH_CPU2, CH_ALU, CH_ALU2, CH_SUB,
  CH_SUB2, CH_BW, CH_BW2, CH_P2, CH_ECALL, CH_CPU3, CH_CPU4, CH_CPU5, CH_CPU6, CH_CPU7, CH_CPU8, CH_QR, CH_DIV, CH_TR, CH_HINT, N_CHIPS
};
/* CPU instance i = 0 .. CPU_INST - 1 <-> chip (the first two keep their old places in the proof order) */
static inline int orc_cpu_chip(int i) { return i == 0 ? CH_CPU : i == 1 ? CH_CPU2 : CH_CPU3 + (i - 2); }
static inline int orc_cpu_instance(int chip) { return chip == CH_CPU ? 0 : chip == CH_CPU2 ? 1 : chip >= CH_CPU3 && chip <= CH_CPU8 ? chip - CH_CPU3 + 2 : -1; }

/* ---- opcodes: Program table column CODE, and the op element of the ALU / sub-word bus tuples ---- */
enum {
  OP_ADD = 1, OP_SUB, OP_XOR, OP_OR, OP_AND, OP_SLL, OP_SRL, OP_SRA, OP_SLT, OP_SLTU, OP_JAL, OP_JALR, OP_BEQ, OP_BNE,
  OP_BLT, OP_BGE, OP_BLTU, OP_BGEU, OP_LB, OP_LH, OP_LW, OP_LBU, OP_LHU, OP_SB, OP_SH, OP_SW, OP_MUL, OP_MULHU,
  OP_ECALL, OP_KECCAK, OP_MULH, OP_MULHSU, OP_DIV, OP_DIVU, OP_REM, OP_REMU, N_OPS_P1
};
/* ---- instruction classes: one selector column each in the CPU row; Program table column CLS ---- */
enum {
  CL_ADD = 1, CL_SUB, CL_ALU, CL_JAL, CL_JALR, CL_BEQ, CL_BNE, CL_BLT, CL_BGE, CL_LW, CL_SW, CL_LDS, CL_STS, CL_ECALL,
  CL_KECCAK, N_CLS_P1
};
#define N_CLS 15
int orc_class_of(uint32_t op);   /* CL_* of an OP_* */
uint32_t orc_code_of(uint32_t op); /* the op a row of that instruction puts on the ALU / sub-word bus (0: none) */
int orc_ucmp_of(uint32_t op);    /* sltu, bltu, bgeu: the unsigned comparison the CPU row does itself (Program column UC) */

/* ---- CPU chip main columns ---- */
enum {
  C_PC = 0, C_TS, C_NEXT_PC,
  C_SEL = 3,                   /* N_CLS class selectors: column C_SEL + (class - 1) */
  C_CODE = C_SEL + N_CLS, C_UC /* unsigned comparison done in this row */, C_WR, C_USE2, C_RD, C_RS1, C_RS2, C_IMM_LO, C_IMM_HI,
  C_TGT_LO, C_TGT_HI,
  C_A,                         /* value written: rd's new value (branches: the taken / less-than flag), a store's new memory word */
  C_B = C_A + 2,               /* reg[rs1] */
  C_C = C_B + 2,               /* second operand: reg[rs2], the immediate, or (loads) the memory word read */
  C_X = C_C + 2,               /* adder output: sum / difference / effective address; sltu, bltu, bgeu: B - C + 2^32 [B < C];
                                  beq, bne: the limb differences' inverses */
  C_K0 = C_X + 2, C_K1,        /* carries / borrows (K1 of an unsigned comparison: B < C); beq, bne: "limb equal" flags */
  C_O1, C_O2, C_O3,            /* byte offset of the effective address 1, 2, 3: at most one is set (offset 0: none); the word
                                  address of a load or store is the linear form X - offset */
  C_ADDR2,                     /* address of the second access: the register rs2, or a load's word address */
  C_ADDR3,                     /* address of the written location: the register rd, or a store's word address */
  C_W_PLO, C_W_PHI,            /* previous value of the written location */
  C_GAP,                       /* 3 access-time differences (rs1; rs2 or load; rd or store): low 16 bits, high 8 bits each; the
                                  previous access time of a slot is its access time - 1 - difference, a linear form */
  CPU_WIDTH = C_GAP + 6
};
/* ---- ecall chip: one row per ecall.  The CPU row of an ecall only moves t0 (reads the code, writes the value left
 *      behind) and hands (time, pc, next pc, code, new t0) over on the ECALL bus; this chip decodes the code into six
 *      flags, reads a0 and a1 itself (memory bus, at the times the CPU row's idle slots would have), sends COMMIT /
 *      COMMIT_DEFERRED words and HALT's exit code to the verifier's buses and decides the next pc ---- */
enum { SC_HALT = 0, SC_WRITE, SC_COMMIT, SC_DEFER, SC_HINT_LEN, SC_HINT_READ };
enum {
  EC_IS_REAL = 0, EC_SC /* 6 flags */, EC_TS = EC_SC + 6, EC_PC, EC_NP, EC_B_LO /* t0: the code */, EC_A_LO, EC_A_HI /* t0 afterwards */,
  EC_C_LO, EC_C_HI /* a0 */, EC_M_LO, EC_M_HI /* a1 */, EC_GAP /* a0, a1: access-time differences, low 16 bits and high 8 */,
  EC_NW = EC_GAP + 4 /* (format v16) a HINT_READ of a1 bytes covers NW = ceil(a1 / 4) words: 4 NW = a1 + P1 + 2 P2; it announces
                        (a0, NW) to the hint chip */, EC_P1, EC_P2,
  ECALL_WIDTH
};

/* ---- keccak chip: p3-keccak-air's 2633 columns (zksp_oracle.h KA_*) + the call time ---- */
#define KC_TS KA_WIDTH
#define KECCAK_WIDTH (KA_WIDTH + 1)

/* ---- keccak-memory chip: 50 rows per call, one state word each ---- */
enum {
  KM_IS_REAL = 0, KM_TS, KM_PTR_LO, KM_PTR_HI, KM_IDX, KM_ISF, KM_ISL, KM_CALL, KM_ADDR, KM_OLD_LO, KM_OLD_HI, KM_NEW_LO,
  KM_NEW_HI, KM_PTS, KM_GL, KM_GH, KMEM_WIDTH
};
/* ---- memory boundary chip: EVERY image address and every other touched address once, strictly increasing ---- */
enum {
  MF_IS_REAL = 0, MF_LO, MF_HI, MF_IS_INIT /* free initial value (not an image address) */, MF_INIT_LO, MF_INIT_HI,
  MF_FIN_LO, MF_FIN_HI, MF_FIN_TS, MF_D_LO, MF_D_HI, MF_BW,
  MF_IS_ZERO /* (format v16) an address outside the image that no HINT_READ covers: it starts as zero.  IS_INIT now means
                "hinted": the initial value comes over the IMG bus from the hint chip, as an image word's comes from the image chip */,
  MEMFINAL_WIDTH
};
/* ---- hint chip (format v16): one row per word of every HINT_READ, in address order within a read.  A read's first row takes
 *      (pointer, number of words) from the ecall chip (HINTR bus); a row the run touches (USED) puts (address, initial value) on
 *      the IMG bus.  Memory outside the image therefore starts with the prover's choice only where a HINT_READ put input ---- */
enum { HN_IS_REAL = 0, HN_FIRST, HN_LAST, HN_ADDR, HN_CNT /* words of the read left, this one included */, HN_LO, HN_HI, HN_USED, HINT_WIDTH };
/* ---- image chip: preprocessed (addr, lo, hi, is_real), main (used = is_real) ---- */
enum { IMG_P_ADDR = 0, IMG_P_LO, IMG_P_HI, IMG_P_REAL, IMAGE_PREP_WIDTH };
#define IMAGE_WIDTH 1
/* ---- program chip: preprocessed instruction fields, main (multiplicity) ---- */
enum { PR_PC = 0, PR_CLS, PR_CODE, PR_UC, PR_WR, PR_USE2, PR_RD, PR_RS1, PR_RS2, PR_IMM_LO, PR_IMM_HI, PR_TGT_LO, PR_TGT_HI, PROGRAM_PREP_WIDTH };
#define PROGRAM_WIDTH 1
/* ---- multiplier chip ---- */
/*      (format v15: mulh / mulhsu as well - the signed high word R follows from the unsigned product's high word:
 *      R + b31 * C + [mulh] c31 * B = P_hi + 2^32 k, limb by limb with carries K0, K1 in {0, 1, 2} as two bits each; R's limbs are
 *      looked up in the range table) */
enum {
  MU_IS_REAL = 0, MU_HI, MU_B, MU_C = MU_B + 32, MU_P = MU_C + 32, MU_Q0 = MU_P + 64, MU_Q1 = MU_Q0 + 10, MU_Q2 = MU_Q1 + 11,
  MU_SH = MU_Q2 + 10 /* mulh */, MU_SHU /* mulhsu */, MU_R /* 2 limbs */, MU_K0 = MU_R + 2 /* 2 bits */, MU_K1 = MU_K0 + 2, MUL_WIDTH = MU_K1 + 2
};
/* ---- divider chip (format v15): div divu rem remu, one row per instruction.  On absolute values |n| = |q| |d| + |r| with
 *      |r| < |d|; the product comes from the multiplier chip over the ALU bus (low word PL, high word zero); signs: q has
 *      sign(n) xor sign(d) unless it is zero, r the sign of n unless it is zero; a zero divisor gives q = 0xffffffff, r = n;
 *      -2^31 / -1 gives q = -2^31, r = 0 by the same relations.  N, D the operands, A the result the CPU row gets ---- */
enum {
  DV_IS_REAL = 0, DV_F /* 4: div divu rem remu */, DV_N = 5, DV_D = 7, DV_A = 9, DV_SN = 11, DV_SD, DV_NH, DV_DH, DV_AN = 15, DV_AD = 17,
  DV_AQ = 19, DV_AR = 21, DV_CN = 23, DV_CD, DV_CQ, DV_CR, DV_Q = 27, DV_R = 29, DV_SQ = 31, DV_SR, DV_XS, DV_PL = 34, DV_K = 36,
  DV_E = 37, DV_BE = 39, DV_NZD, DV_INVD, DV_NZQ, DV_INVQ, DV_NZR, DV_INVR, DIV_WIDTH
};
/* ---- ALU chip: sll srl sra and the signed slt (slt, blt, bge) over bits ---- */
enum {
  AL_IS_REAL = 0, AL_SEL /* 4 selectors, OP_SLL..OP_SLT */, AL_A = AL_SEL + 4, AL_B = AL_A + 2, AL_C = AL_B + 32,
  AL_X = AL_C + 32 /* one-hot shift amount / comparison difference */, AL_K0 = AL_X + 32, AL_K1, ALU_WIDTH
};
/* ---- bitwise chip: xor or and byte by byte, every (b, c, a) byte triple looked up in the table chip ---- */
enum { BW_IS_REAL = 0, BW_SEL /* 3 selectors: XOR OR AND */, BW_A = BW_SEL + 3, BW_B = BW_A + 4, BW_C = BW_B + 4, BW_WIDTH = BW_C + 4 };
/* ---- sub-word chip: lb lh lbu lhu sb sh; the memory word and the stored limb as bytes (range-checked through the table
 *      chip's byte-pair rows), the sign bit a signed load extends bound to its byte by a byte-operation lookup
 *      (byte AND 0x80 = 128 * sign) ---- */
enum {
  SW_IS_REAL = 0, SW_SEL /* 6 selectors: LB LH LBU LHU SB SH */, SW_O = SW_SEL + 6 /* 4: byte offset, one-hot */,
  SW_A = SW_O + 4 /* loads: the value loaded; stores: the word left behind */, SW_MB = SW_A + 2 /* the memory word's 4 bytes (stores: before) */,
  SW_CB = SW_MB + 4 /* the 2 bytes of the stored register's low limb */,
  SW_S = SW_CB + 2 /* sign bit a signed load extends */, SW_SELB /* the byte that carries it */,
  SUB_WIDTH = SW_SELB + 1
};
/* ---- Poseidon2 chip (row f4): one width-16 permutation per row.  Six kinds of rows:
 *   N   (stage 1) a node K of a binary heap of digests (root 1, children 2K and 2K + 1; K as two range-checked limbs): consumes
 *       its children's digests from the DIGEST bus and produces its own; the verifier supplies digests at keys of its choice
 *       and takes the root (all n leaves of a tree: the aggregation root; or one leaf and the siblings along its path);
 *   SZ / SC  (stage 2a: the openings of a leaf proof) a sponge row: eight absorbed words over the capacity carried from the
 *       row before (SC) or over the zero state (SZ: the first block of a hash; the last block of an input is zero-filled);
 *   PL / PR  a step of a Merkle path: the running digest (the output of the row before) is the left / right input, the
 *       sibling the other, free; K' = 2 K + [right], M' = 2 M;
 *   J   an injection of a mixed-height tree: the running digest on the left, on the right the hash of the shorter matrices'
 *       row, consumed from the DIGEST bus where the sponge that made it (a segment of sponge rows elsewhere, labelled with
 *       this row's T, K, M) put it; K' = K, M' = M + 1.
 * A run (one opening) starts with a sponge over the zero state flagged NEW (K = 1, M = 0): K collects the position bits, M
 * the levels an injection followed, T names the opening; its last row sends (T, 0, K, M, digest), which the VERIFIER
 * consumes with the root it knows - position, shape and root of every opening are the verifier's.  Columns: the labels and
 * flags, the input state, and per S-box its cube and its seventh power (degree <= 3). ---- */
enum {
  P2_IS_REAL = 0, P2_KL, P2_KH /* the key / position accumulator, two limbs */, P2_T /* tag of the opening */, P2_M /* injection mask */,
  P2_FN, P2_SZ, P2_SC, P2_PL, P2_PR, P2_FJ /* row kind, one-hot on real rows */, P2_NEW /* (sponge rows) the run's first hash */,
  P2_SND /* (sponge rows) the hash goes to an injection row */, P2_FR /* a FRI leaf: the absorbed pair goes to the query chip */,
  P2_IN /* 16 */, P2_EXT = P2_IN + 16 /* 8 external rounds x (16 cubes, 16 outputs) */,
  P2_INT = P2_EXT + 256 /* 13 internal rounds x (cube, output) */,
  /* format v16 (stage 2b): RE - the last row of a run: its digest is compared with the root RID names (ROOT bus: the transcript
   * chip's, or the verifier's for the preprocessed tree), its position goes to the query chip (POS bus); SE - the last block of
   * a matrix row's hash: the Horner sum SO (in alpha_f, over every absorbed word, block by block:
   * SO' = SO alpha^8 + sum_{i<8} alpha^(7-i) in_i) goes to the query chip (SEG bus); AP: alpha_f^1 .. alpha_f^8 */
  P2_RE = P2_INT + 26, P2_SE, P2_RID, P2_SO /* 4 */, P2_AP = P2_SO + 4 /* 8 x 4 */, P2CHIP_WIDTH = P2_AP + 32
};
/* row records (P2_REC_WORDS each): flags = kind | P2F_*, tag, key, mask, the 16 input words, RID, SO (4), alpha_f (4), padding */
enum { P2K_NONE = 0, P2K_NODE, P2K_SZ, P2K_SC, P2K_PL, P2K_PR, P2K_J };
#define P2_REC_WORDS 32
#define P2_REC_RID 20
#define P2_REC_SO 21
#define P2_REC_ALPHA 25
#define P2F_NEW 16u
#define P2F_SND 32u
#define P2F_FRI 64u
#define P2F_RE 128u
#define P2F_SE 256u
/* tags of a leaf proof's openings: leaf l (its place among the leaves checked beside one run), query q, tree r (0
 * preprocessed, 1 main, 2 permutation, 3 quotient, 4 + k: FRI layer k); ids of its commitment roots */
#define LEAF_TAG_STRIDE 64u
#define LEAF_TAG_LEAF_STRIDE (1u << 18)
#define LEAF_TAG(l, q, r) (1u + LEAF_TAG_STRIDE * (q) + LEAF_TAG_LEAF_STRIDE * (l) + (r))
#define LEAF_RID(l, r) (64u * (l) + (r))
/* ---- query chip (row f4, stage 2b; it stands where stage 2a's fold chip stood): 31 rows per query of a checked leaf proof,
 *      one per bit of the word the leaf's transcript drew for the query, from bit 30 down (J).  The bits are the canonical
 *      decomposition of the word (ACC from the top; EQ: the bits so far equal those of p - 1 = 0x78000000; F1..F3 mark the rows of
 *      bits 29..27).  Bit lm is the coset (row CSR), the bits below the position; the rows of the bits lm - 1 .. 0 are the FRI
 *      layers K = 0 .. lm - 1 (LAY; FL: layer 0).  POW = 2^J, LOW = the word mod 2^J, REV = its low J bits reversed: the keys of
 *      the openings are linear in them.  A layer row receives its layer's opening (POS, PAIR, BETA), folds (E, F, XINV), and -
 *      where a height joins (HASRO; HAS0: with preprocessed columns) - forms the reduced opening RO of that height from the
 *      Horner sums H0..H3 of the four trees' opened rows (SEG), zeta, alpha_f / delta and the verifier's constants B1, B2, w_H
 *      (BCONST).  KEYJ / MJ (KEY0 / M0 for the 2^16-tall preprocessed tree, whose opening is received on the row of bit 16: PR0):
 *      key and mask of the injected rows; MT / MT0: the masks of the whole runs.  The domain point: OMI = 1 / omega (order
 *      2^(lm+1), the verifier's), R = omega^-(the position bits so far) by square and multiply, YT = 1 / y for the tallest
 *      height, YKI its squares down the layers, GI = g^-(2^K). ---- */
enum {
  QR_IS_REAL = 0, QR_FIRST, QR_LAST, QR_LEAF, QR_QL, QR_J, QR_BIT, QR_ACC, QR_EQ, QR_F1, QR_F2, QR_F3, QR_CSR, QR_FL, QR_LAY, QR_K,
  QR_CS, QR_POW, QR_LOW, QR_REV, QR_PR0, QR_CNT0, QR_KEYJ, QR_MJ, QR_MT, QR_P0A, QR_KEY0, QR_M0, QR_MT0, QR_HASRO, QR_HAS0, QR_OMI,
  QR_MU, QR_CSM, QR_R, QR_R2, QR_YT, QR_YKI, QR_GI, QR_XINV, QR_WH,
  QR_BETA = 41, QR_LO = 45, QR_HI = 49, QR_E = 53, QR_F = 57, QR_RO = 61, QR_H = 65 /* 4 x 4 */, QR_AF = 81, QR_DL = 85, QR_D2 = 89,
  QR_D3 = 93, QR_D4 = 97, QR_G2 = 101, QR_ZETA = 105, QR_ZW = 109, QR_D0 = 113, QR_D1 = 117, QR_B1 = 121, QR_B2 = 125, QR_WIDTH = 129
};
#define QR_REC_WORDS 132 /* a row record is the row: QR_WIDTH canonical words, then padding */
#define F_GEN_INV 64944062u /* 1 / F_GEN mod p */
/* ---- transcript chip (row f4, stage 2b): one duplex of a checked leaf proof's Fiat-Shamir transcript per row - "absorb eight
 *      words" (ABS) or "squeeze" (since format v16 every phase of a transcript ends on a block boundary).  The verifier
 *      dictates every row (TBLK / TSQ: what it absorbs, what its inputs and outputs are used for: the U flags, RIDK, QBASE); the
 *      rows hand the commitment roots (ROOT), the final constant (FINAL), zeta, alpha_f / delta and the FRI betas to the chips
 *      that check the queries, every query's index word to the query chip (QIDX: output word 7 - j is query QBASE + j), and the
 *      proof-of-work word to the verifier. ---- */
enum {
  TR_IS_REAL = 0, TR_LEAF, TR_STEP, TR_FIRST, TR_ABS, TR_UROOT, TR_UZETA, TR_UAF, TR_UBETA, TR_UFIN, TR_UPOW, TR_UQ, TR_QM /* 8 */,
  TR_RIDK = TR_QM + 8, TR_QBASE, TR_MROOT, TR_MFIN, TR_MZETA, TR_MAF, TR_MBETA, TR_IN /* 16 */, TR_EXT = TR_IN + 16, TR_INT = TR_EXT + 256,
  TR_WIDTH = TR_INT + 26
};
/* row records (TR_REC_WORDS each): word 0 = the flags word of the verifier's tuple (UROOT + 2 UZETA + 4 UAF + 8 UBETA + 16 UFIN
 * + 32 UPOW + 64 UQ + 128 * the QM bits) | FIRST << 16 | ABS << 17; leaf, step, RIDK, QBASE, the five multiplicities, the 16 input words */
#define TR_REC_WORDS 32
/* ---- table chip: 2^16 rows; preprocessed (x = low byte, y = high byte, na = row index not a multiple of 4,
 *      nt = row index zero or above ADDR_HI_MAX, x ^ y, x & y); main: multiplicities of range16 (kind 0), 4-aligned range16
 *      (kind 1), high address limb (kind 2: 1 .. ADDR_HI_MAX), byte pair, and the byte operations xor / or / and ---- */
enum { TB_P_X = 0, TB_P_Y, TB_P_NA, TB_P_NT, TB_P_XOR, TB_P_AND, TABLE_PREP_WIDTH };
enum { TB_M_R16 = 0, TB_M_AL, TB_M_TOP, TB_M_BY, TB_M_XOR, TB_M_OR, TB_M_AND, TABLE_WIDTH };
#define TABLE_LOG_H 16
#define ADDR_HI_MAX 0x77FEu /* high limb of the largest address / jump target: values stay below 0x78000000 < p; the
                               smallest is 1: no load, store or keccak state can name a register (addresses 0 .. 31) */

/* ---- buses ---- */
enum { BUS_MEM = 1, BUS_PROG, BUS_KCALL, BUS_KIO, BUS_ALU, BUS_PUBC, BUS_PUBH, BUS_RANGE, BUS_BYTES, BUS_SUB, BUS_IMG, BUS_BYTEOP, BUS_DIGEST, BUS_ECALL, BUS_PAIR,
       BUS_POS, BUS_ROOT, BUS_SEG, BUS_TBLK, BUS_TSQ, BUS_FINAL, BUS_ZETA, BUS_AF, BUS_BETA, BUS_POW, BUS_QIDX, BUS_LEAFK, BUS_BCONST, BUS_HINTR };

/* A linear form over the row [preprocessed | main]: c0 + sum coef[i] * row[col[i]] (canonical words). */
#define LF_MAX 40
typedef struct { int n; int col[LF_MAX]; uint32_t coef[LF_MAX]; uint32_t c0; } orc_lf;
#define INTER_MAX_ELEMS 13
typedef struct { int bus; int sign; /* +1 send / produce, -1 receive / consume */ orc_lf mult; int n_el; orc_lf el[INTER_MAX_ELEMS]; } orc_inter;

typedef struct {
  const char* name;
  int prep_width, main_width;
  int n_inter;
  const orc_inter* inter;
  int n_constraints;  /* base-field constraints; the LogUp constraints follow */
  int n_merged;       /* the last n_merged interactions are sends with boolean, mutually exclusive multiplicities (one
                         instruction class each): they share ONE fraction, (sum m_k) / (sum m_k f_k + 1 - sum m_k) */
} orc_chip;
const orc_chip* orc_machine_chip(int chip);
/* LogUp layout.  The interactions before the merged ones are taken two at a time; every such pair (or last single one),
 * and the merged group, is a SLOT whose value at a row is the sum of its fractions.  All slots but the last have a helper
 * column (an extension element, 4 base columns) constrained to the slot's value; the last slot has none: its value is
 * phi_next - phi + cum / H - (sum of the helper columns), where phi is the running-sum column, cum the chip's cumulative
 * sum (a proof word) and H the height, on EVERY row, cyclically (no boundary rows: summing the rows gives
 * cum = sum of all fractions).  Every LogUp constraint has degree <= 3. */
/* Quotients.  The chips of one height share ONE quotient: with total_c = n_constraints + slots the number of constraints of
 * chip c (base constraints, then one LogUp constraint per slot), chip c's constraints are folded with the powers
 * alpha^(off_c + k), off_c = the sum of total_c' over the chips c' < c of the same height, and the sum over the height's
 * chips is divided by the vanishing polynomial once.  The first chip of a height (its "leader") carries the quotient's
 * 8 columns (two chunks of one extension element each); the others have none. */
static inline int orc_quot_leader(const int* logh, int c) {
  for (int c2 = 0; c2 < c; ++c2)
    if (logh[c2] == logh[c]) return c2;
  return c;
}
static inline int orc_chip_slots(const orc_chip* c);
static inline int orc_quot_alpha_offset(const int* logh, int c) {
  int off = 0;
  for (int c2 = 0; c2 < c; ++c2)
    if (logh[c2] == logh[c]) off += orc_machine_chip(c2)->n_constraints + orc_chip_slots(orc_machine_chip(c2));
  return off;
}
static inline int orc_chip_slots(const orc_chip* c) { return (c->n_inter - c->n_merged + 1) / 2 + (c->n_merged ? 1 : 0); }
static inline int orc_chip_helpers(const orc_chip* c) { return orc_chip_slots(c) - 1; }
static inline int orc_chip_perm_width(const orc_chip* c) { return 4 * orc_chip_slots(c); }

/* ---- inputs: exactly the arrays zksp_mtrace_section() exposes ---- */
typedef struct {
  const uint32_t* program; size_t n_program;   /* 9 u32 per row; the last row is the padding instruction (jal to itself) */
  const uint32_t* image; size_t n_image;       /* 2 u32 per row */
  uint32_t entry, text_base;
  int log_prog, log_image;
  const uint32_t* cycles; size_t n_cycles;     /* 12 u32 per cycle */
  const uint8_t* keccak; size_t n_keccak;      /* 408 bytes per call */
  const uint32_t* memfinal; size_t n_memfinal; /* 5 u32: addr, init, fin, fin_ts, is_init (0 image, 1 hinted, 2 starts as zero); row 0 is x0, closed at its last real access */
  const uint32_t* muls; size_t n_muls;         /* 3 u32: kind (0 mul, 1 mulhu, 2 mulh, 3 mulhsu), b, c */
  const uint32_t* prog_mult;                   /* n_program; the padding row holds 0 (its fetches depend on the heights) */
  const int* shape;                            /* NULL: the minimal heights; else N_CHIPS log heights the run fits (a batch of
                                                  runs is proven with one shape: the heights of their largest counts) */
  const uint32_t* agg_keys;                    /* heap keys of the payload's digests (NULL: n_agg + j, the leaves of a full tree) */
  const uint32_t* agg_leaves; size_t n_agg;    /* aggregation payload: n_agg (0, or a power of two >= 2) digests of 8 words
                                                  whose Poseidon2 Merkle root the proof also establishes */
  /* leaf-proof check (row f4, stage 2a): the records zksp_mtrace_section() exposes - the Poseidon2-chip rows after the
   * aggregation payload's node rows, the fold-chip rows, and the public bus tuples (PUB_TUPLE_WORDS each: bus, 1 = the
   * verifier sends it / 0 = receives it, multiplicity, number of elements, 12 element slots) that state what was checked */
  const uint32_t* leaf_p2_rows; size_t n_leaf_p2;
  const uint32_t* leaf_qr_rows; size_t n_leaf_qr;   /* query chip rows (QR_REC_WORDS each) */
  const uint32_t* leaf_tr_rows; size_t n_leaf_tr;   /* transcript chip rows (TR_REC_WORDS each) */
  const uint32_t* pub_tuples; size_t n_pub;
} orc_machine_input;
#define PUB_TUPLE_WORDS 16

/* The oracle's own event lists (cycle indices of the ALU-chip and sub-word-chip rows, in execution order) and the
 * last access time of x0 by a real cycle; the product's tracer emits the same lists and tests compare them. */
size_t orc_machine_events(const orc_machine_input* in, int which /* 0 alu, 1 sub-word, 2 bitwise, 3 ecall, 4 divider */, uint32_t* out /* may be NULL */);
uint32_t orc_machine_x0_last(const orc_machine_input* in);

/* log2 trace height of every chip for this input (minimum 5) */
void orc_machine_heights(const orc_machine_input* in, int logh[N_CHIPS]);
/* column-major traces: prep [prep_width][H] (NULL when the chip has none), main [main_width][H] */
void orc_machine_fill(const orc_machine_input* in, int chip, int logh, uint32_t* prep, uint32_t* main_);
/* Public scalars of a CPU instance: pc and time of its first row, whether another instance continues it, the pc
 * that one starts at (the hand-over pc, a proof-header word the transcript absorbs), and the padding pc. */
enum { CPUPUB_START_PC = 0, CPUPUB_START_TS, CPUPUB_HAS_SUCC, CPUPUB_END_PC, CPUPUB_PAD_PC, CPUPUB_N };
void orc_machine_cpu_pub(const orc_machine_input* in, int chip, uint32_t pub[CPUPUB_N]);
/* base constraints of one chip at one row pair; `pub` = the CPUPUB_* words for the CPU instances (ignored elsewhere) */
void orc_machine_constraints(int chip, const uint32_t* prep, const uint32_t* loc, const uint32_t* nxt, uint32_t is_first,
                             uint32_t is_last, uint32_t is_trans, const uint32_t* pub, uint32_t* out);

/* ---- kernel-level parity: one chip's stages for given challenges (4 canonical words each) ---- */
/* LogUp permutation trace [perm_width][H] (helper columns, then the running sum phi, phi_0 = 0) and the chip's cumulative sum */
void orc_machine_stage_perm(const orc_machine_input* in, int chip, const uint32_t gamma[4], const uint32_t beta[4], uint32_t* perm,
                            uint32_t cum[4]);
/* this chip's share of its height's quotient, [8][H]: columns 4c..4c+3 = the extension element over coset c; the
 * quotient the height's first chip commits is the sum of the shares of the chips of that height */
void orc_machine_stage_quotient(const orc_machine_input* in, int chip, const uint32_t alpha[4], const uint32_t gamma[4],
                                const uint32_t beta[4], uint32_t* quot);

/* ---- whole machine proof ---- */
typedef struct {
  uint32_t exit_code;
  uint32_t pv_len;
  uint32_t pv_digest[8];
  uint32_t deferred_digest[8];
} orc_machine_public;
/* the aggregation payload's public part: Merkle root of the leaves (2-to-1 Poseidon2 compressions) and the sponge hash
 * of the leaf list, which stands for the list in the transcript */
void orc_machine_agg_public(const uint32_t* leaves, size_t n, uint32_t root[8], uint32_t list_digest[8]);
/* the same for digests supplied at heap keys (keys NULL: n + j): the root is node 1's digest, the list digest covers keys and
 * digests.  Returns 0 if the supplied set is malformed (a repeated or out-of-range key, an ancestor that is itself supplied
 * or lacks a child). */
int orc_machine_nodes_public(const uint32_t* keys, const uint32_t* digests, size_t n, uint32_t root[8], uint32_t list_digest[8]);
/* the rows of the Poseidon2 chip for a supplied set: every ancestor of a supplied key in ascending order, 25 words each
 * (key, left child's digest, right child's digest, own digest).  Returns the number of rows, (size_t)-1 if malformed;
 * rows may be NULL. */
size_t orc_machine_agg_rows(const uint32_t* keys, const uint32_t* digests, size_t n, uint32_t* rows);
#define ZKSP_VERSION_MACHINE 16u
/* vk: preprocessed commitment root + digest binding entry pc, table heights and keccak mode */
void orc_machine_setup(const orc_machine_input* in, int keccak_mode, uint32_t prep_root[8], uint32_t vk_digest[8]);
/* coefficients of the reduced openings, one per opened value (mprover.c) */
void orc_reduce_coefs(const int logh[N_CHIPS], fe4 af, fe4 delta, fe4* out);
size_t orc_machine_proof_size(const int logh[N_CHIPS], int log_prog, int log_image, const orc_config* cfg, uint32_t pv_len);
int orc_machine_prove(const orc_machine_input* in, int keccak_mode, const orc_machine_public* pub, const uint8_t* public_values,
                      const orc_config* cfg, uint8_t* out, size_t cap, size_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
