/* ORACLE -- TEST INFRASTRUCTURE ONLY (see field.h and machine.h).
 *
 * Chips of the machine proof (format v12): bus interactions, trace generation from the executor's
 * records and base-field constraints.  This repository's own arithmetisation (machine.h header
 * note); what it must reproduce is the reference's statement: the committed RV32IM guest
 * (circuits/sp1-merkle-proof/src/main.rs:4-14 running crypto-ops/src/lib.rs:8-23) executed from its
 * entry point to HALT with the committed public values.  PARITY UNPINNED vs sp1-core-machine 3.4.0
 * (reference Cargo.lock:7130; sources absent).
 *
 * Design in one paragraph.  One CPU row per cycle and every row is an instruction: after HALT the
 * rows execute the padding instruction the Program table ends with (a jump to itself), so the
 * instruction fetch - a lookup into the preprocessed Program table - vouches for every decoded
 * field of every row.  Operands are 16-bit limbs.  The row adds, subtracts, tests equality, moves
 * words and forms addresses; xor / or / and / shifts / less-than go to the ALU chip, sub-word loads
 * and stores to the sub-word chip, mul / mulhu to the multiplier (one row per such instruction,
 * bits inside), keccak-f calls to keccak-memory + keccak.  Registers are memory addresses 0..31.
 * Memory consistency is the offline argument: every access consumes the tuple (addr, value, time)
 * its predecessor produced and produces its own; the memory-boundary chip lists EVERY address of
 * the program image and every other touched address exactly once, in strictly increasing order
 * (compared limb-wise over the integers), opens each with its initial value (image addresses: by
 * lookup into the preprocessed Image table, which sends every word exactly once; others: a free,
 * range-checked value - SP1's treatment of uninitialised / hinted memory) and closes it with its
 * final one.  "previous time < time": the difference is a 16-bit and an 8-bit limb looked up in the
 * table chip (2^16 rows: range16, 4-aligned range16, byte pairs).  COMMIT and HALT post the public
 * digest words and the exit code on buses the verifier closes.
 */
#include <stdlib.h>
#include <string.h>

#include "machine.h"

/* ------------------------------------------------------------------------------------------
 * opcodes and classes
 * ---------------------------------------------------------------------------------------- */
int orc_class_of(uint32_t op) {
  switch (op) {
    case OP_ADD: return CL_ADD;
    case OP_SUB: return CL_SUB;
    case OP_XOR: case OP_OR: case OP_AND: case OP_SLL: case OP_SRL: case OP_SRA: case OP_SLT: case OP_SLTU:
    case OP_MUL: case OP_MULHU: case OP_MULH: case OP_MULHSU: case OP_DIV: case OP_DIVU: case OP_REM: case OP_REMU: return CL_ALU;
    case OP_JAL: return CL_JAL;
    case OP_JALR: return CL_JALR;
    case OP_BEQ: return CL_BEQ;
    case OP_BNE: return CL_BNE;
    case OP_BLT: case OP_BLTU: return CL_BLT;
    case OP_BGE: case OP_BGEU: return CL_BGE;
    case OP_LW: return CL_LW;
    case OP_SW: return CL_SW;
    case OP_LB: case OP_LH: case OP_LBU: case OP_LHU: return CL_LDS;
    case OP_SB: case OP_SH: return CL_STS;
    case OP_ECALL: return CL_ECALL;
    case OP_KECCAK: return CL_KECCAK;
    default: return 0;
  }
}
uint32_t orc_code_of(uint32_t op) {
  switch (op) {
    case OP_XOR: case OP_OR: case OP_AND: case OP_SLL: case OP_SRL: case OP_SRA: case OP_SLT: case OP_SLTU:
    case OP_MUL: case OP_MULHU: case OP_MULH: case OP_MULHSU: case OP_DIV: case OP_DIVU: case OP_REM: case OP_REMU:
    case OP_LB: case OP_LH: case OP_LBU: case OP_LHU: case OP_SB: case OP_SH: return op;
    case OP_BLT: case OP_BGE: return OP_SLT;
    case OP_BLTU: case OP_BGEU: return OP_SLTU;
    default: return 0;
  }
}
int orc_ucmp_of(uint32_t op) { return op == OP_SLTU || op == OP_BLTU || op == OP_BGEU; }
/* does a cycle of this op occupy a row of the ALU chip (0) / the sub-word chip (1) / the bitwise chip (2)? */
static int event_kind(uint32_t op) {
  const uint32_t code = orc_code_of(op);
  if (code >= OP_SLL && code <= OP_SLT) return 0;
  if (code == OP_LB || code == OP_LH || code == OP_LBU || code == OP_LHU || code == OP_SB || code == OP_SH) return 1;
  if (code >= OP_XOR && code <= OP_AND) return 2;
  if (op == OP_ECALL) return 3;
  if (op >= OP_DIV && op <= OP_REMU) return 4; /* the divider chip */
  return -1;
}

/* ------------------------------------------------------------------------------------------
 * linear forms and interactions
 * ---------------------------------------------------------------------------------------- */
static void lf_zero(orc_lf* f) { f->n = 0; f->c0 = 0; }
static void lf_add(orc_lf* f, int col, uint32_t coef) {
  if (f->n >= LF_MAX) abort();
  f->col[f->n] = col;
  f->coef[f->n] = coef % FP;
  f->n++;
}
static orc_lf lf_col(int col) { orc_lf f; lf_zero(&f); lf_add(&f, col, 1); return f; }
static orc_lf lf_const(uint32_t c) { orc_lf f; lf_zero(&f); f.c0 = c % FP; return f; }
static orc_lf lf_plus(orc_lf f, uint32_t c) { f.c0 = f_add(f.c0, c % FP); return f; }
/* a + k * b */
static orc_lf lf_pair(int a, int b, uint32_t k) { orc_lf f; lf_zero(&f); lf_add(&f, a, 1); lf_add(&f, b, k); return f; }
/* sum of the given columns */
static orc_lf lf_sum(const int* cols, int n) { orc_lf f; lf_zero(&f); for (int i = 0; i < n; ++i) lf_add(&f, cols[i], 1); return f; }
/* c - col */
static orc_lf lf_const_minus(uint32_t c, int col) { orc_lf f; lf_zero(&f); lf_add(&f, col, FP - 1); f.c0 = c % FP; return f; }
/* n-bit little-endian value of the bit columns starting at `bits` */
static orc_lf lf_bits(int bits, int n) {
  orc_lf f;
  lf_zero(&f);
  for (int i = 0; i < n; ++i) lf_add(&f, bits + i, 1u << i);
  return f;
}
#define SELC(cls) (C_SEL + (cls) - 1)

#define CPU_INTER 19
static orc_inter g_cpu[CPU_INTER], g_keccak[50], g_kmem[8], g_memfinal[10], g_image[1], g_program[1], g_mul[5], g_div[13], g_table[7],
    g_alu[1], g_sub[5], g_bw[5], g_p2[10], g_ecall[12], g_qr[17], g_tr[16], g_hint[2];
static orc_chip g_chips[N_CHIPS];
static int g_ready = 0;

static orc_inter mem_inter(int sign, orc_lf mult, orc_lf addr, orc_lf lo, orc_lf hi, orc_lf ts) {
  orc_inter it;
  memset(&it, 0, sizeof it);
  it.bus = BUS_MEM; it.sign = sign; it.mult = mult; it.n_el = 4;
  it.el[0] = addr; it.el[1] = lo; it.el[2] = hi; it.el[3] = ts;
  return it;
}
/* table lookups: range16 (kind 0: any 16-bit value; kind 1: a multiple of 4) and byte pairs */
static orc_inter range_inter(int sign, orc_lf mult, orc_lf kind, orc_lf value) {
  orc_inter it;
  memset(&it, 0, sizeof it);
  it.bus = BUS_RANGE; it.sign = sign; it.mult = mult; it.n_el = 2;
  it.el[0] = kind; it.el[1] = value;
  return it;
}
static orc_inter bytes_inter(int sign, orc_lf mult, orc_lf x, orc_lf y) {
  orc_inter it;
  memset(&it, 0, sizeof it);
  it.bus = BUS_BYTES; it.sign = sign; it.mult = mult; it.n_el = 2;
  it.el[0] = x; it.el[1] = y;
  return it;
}

static int count_constraints(int chip);

static void build(void) {
  if (g_ready) return;
  const orc_lf one = lf_const(1), zero = lf_const(0), ts = lf_col(C_TS);
  const orc_lf a_lo = lf_col(C_A), a_hi = lf_col(C_A + 1), b_lo = lf_col(C_B), b_hi = lf_col(C_B + 1), c_lo = lf_col(C_C),
               c_hi = lf_col(C_C + 1);
  /* ---- CPU ---- */
  {
    orc_inter* it = &g_cpu[0];
    memset(it, 0, sizeof *it);
    it->bus = BUS_PROG; it->sign = -1; it->mult = one; it->n_el = 13;
    it->el[0] = lf_col(C_PC);
    lf_zero(&it->el[1]);
    for (int k = 1; k <= N_CLS; ++k) lf_add(&it->el[1], SELC(k), (uint32_t)k);
    it->el[2] = lf_col(C_CODE); it->el[3] = lf_col(C_UC); it->el[4] = lf_col(C_WR); it->el[5] = lf_col(C_USE2); it->el[6] = lf_col(C_RD);
    it->el[7] = lf_col(C_RS1); it->el[8] = lf_col(C_RS2); it->el[9] = lf_col(C_IMM_LO); it->el[10] = lf_col(C_IMM_HI);
    it->el[11] = lf_col(C_TGT_LO); it->el[12] = lf_col(C_TGT_HI);
  }
  /* Three accesses per row: rs1 at ts, the second operand (rs2, or a load's memory word) at ts + 1, the written location
   * (rd, or a store's memory word) at ts + 2.  Previous access time of slot q: ts + q - 1 - (gap_lo + 2^16 gap_hi). */
  orc_lf pts[3];
  for (int q = 0; q < 3; ++q) {
    lf_zero(&pts[q]);
    lf_add(&pts[q], C_TS, 1); lf_add(&pts[q], C_GAP + 2 * q, FP - 1); lf_add(&pts[q], C_GAP + 2 * q + 1, FP - 65536);
    pts[q].c0 = f_sub((uint32_t)q, 1);
  }
  g_cpu[1] = mem_inter(-1, one, lf_col(C_RS1), b_lo, b_hi, pts[0]);
  g_cpu[2] = mem_inter(+1, one, lf_col(C_RS1), b_lo, b_hi, ts);
  {
    /* rs2 (USE2) or a load: exclusive by the Program table (a load's second operand is its immediate) */
    const int m2_c[3] = {C_USE2, SELC(CL_LW), SELC(CL_LDS)};
    const orc_lf m2 = lf_sum(m2_c, 3);
    g_cpu[3] = mem_inter(-1, m2, lf_col(C_ADDR2), c_lo, c_hi, pts[1]);
    g_cpu[4] = mem_inter(+1, m2, lf_col(C_ADDR2), c_lo, c_hi, lf_plus(ts, 1));
    /* rd (WR) or a store (which writes no register) */
    const int m3_c[3] = {C_WR, SELC(CL_SW), SELC(CL_STS)};
    const orc_lf m3 = lf_sum(m3_c, 3);
    g_cpu[5] = mem_inter(-1, m3, lf_col(C_ADDR3), lf_col(C_W_PLO), lf_col(C_W_PHI), pts[2]);
    g_cpu[6] = mem_inter(+1, m3, lf_col(C_ADDR3), a_lo, a_hi, lf_plus(ts, 2));
  }
  /* access-time differences: every row looks up its three low limbs and the high bytes (zero where the access is not
   * live) */
  for (int q = 0; q < 3; ++q) g_cpu[7 + q] = range_inter(-1, one, zero, lf_col(C_GAP + 2 * q));
  g_cpu[10] = bytes_inter(-1, one, lf_col(C_GAP + 1), lf_col(C_GAP + 3));
  g_cpu[11] = bytes_inter(-1, one, lf_col(C_GAP + 5), zero);
  {
    /* the adder output is canonical, an address is word-aligned once its byte offset is taken off, and addresses,
     * jump targets and the keccak call's return address stay below 0x78000000 (their high limb is looked up as kind 2) */
    /* ... and so is the difference of an unsigned comparison (UC) */
    const int chk_c[10] = {SELC(CL_ADD), SELC(CL_SUB), SELC(CL_JALR), SELC(CL_LW), SELC(CL_SW), SELC(CL_LDS), SELC(CL_STS),
                           SELC(CL_ECALL), SELC(CL_KECCAK), C_UC};
    const int al_c[5] = {SELC(CL_JALR), SELC(CL_LW), SELC(CL_SW), SELC(CL_LDS), SELC(CL_STS)};
    const int top_c[6] = {SELC(CL_JALR), SELC(CL_LW), SELC(CL_SW), SELC(CL_LDS), SELC(CL_STS), SELC(CL_KECCAK)};
    const orc_lf chk = lf_sum(chk_c, 10);
    orc_lf xoff = lf_col(C_X);
    lf_add(&xoff, C_O1, FP - 1); lf_add(&xoff, C_O2, FP - 2); lf_add(&xoff, C_O3, FP - 3);
    orc_lf top = lf_sum(top_c, 6);
    for (int i = 0; i < top.n; ++i) top.coef[i] = 2; /* kind 2: the high limb of an address is in 1 .. ADDR_HI_MAX */
    g_cpu[12] = range_inter(-1, chk, top, lf_col(C_X + 1));
    g_cpu[13] = range_inter(-1, chk, lf_sum(al_c, 5), xoff);
  }
  {
    orc_inter* it = &g_cpu[14];
    memset(it, 0, sizeof *it);
    /* every ALU-class instruction and every ordered branch, except the unsigned comparisons the row does itself */
    const int alu_c[3] = {SELC(CL_ALU), SELC(CL_BLT), SELC(CL_BGE)};
    it->bus = BUS_ALU; it->sign = +1; it->mult = lf_sum(alu_c, 3); lf_add(&it->mult, C_UC, FP - 1); it->n_el = 7;
    it->el[0] = lf_col(C_CODE);
    it->el[1] = a_lo; it->el[2] = a_hi; it->el[3] = b_lo; it->el[4] = b_hi; it->el[5] = c_lo; it->el[6] = c_hi;
    /* sub-word loads: (op, offset, value loaded, memory word, 0); sub-word stores: (op, offset, word left behind, word
     * before, low limb of the stored register) - one tuple layout, two sends because the word sits in other columns */
    for (int st = 0; st < 2; ++st) {
      it = &g_cpu[15 + st];
      memset(it, 0, sizeof *it);
      it->bus = BUS_SUB; it->sign = +1; it->mult = lf_col(SELC(st ? CL_STS : CL_LDS)); it->n_el = 7;
      it->el[0] = lf_col(C_CODE);
      lf_zero(&it->el[1]); lf_add(&it->el[1], C_O1, 1); lf_add(&it->el[1], C_O2, 2); lf_add(&it->el[1], C_O3, 3);
      it->el[2] = a_lo; it->el[3] = a_hi;
      it->el[4] = st ? lf_col(C_W_PLO) : c_lo; it->el[5] = st ? lf_col(C_W_PHI) : c_hi;
      it->el[6] = st ? c_lo : zero;
    }
    it = &g_cpu[17];
    memset(it, 0, sizeof *it);
    it->bus = BUS_KCALL; it->sign = +1; it->mult = lf_col(SELC(CL_KECCAK)); it->n_el = 3;
    it->el[0] = ts; it->el[1] = c_lo; it->el[2] = c_hi;
    it = &g_cpu[18];
    memset(it, 0, sizeof *it);
    /* an ecall: the ecall chip takes it from here (time, pc, next pc, the code in t0, the value left in t0) */
    it->bus = BUS_ECALL; it->sign = +1; it->mult = lf_col(SELC(CL_ECALL)); it->n_el = 6;
    it->el[0] = ts; it->el[1] = lf_col(C_PC); it->el[2] = lf_col(C_NEXT_PC); it->el[3] = b_lo; it->el[4] = a_lo; it->el[5] = a_hi;
  }
  /* ---- ecall chip ---- */
  {
    const orc_lf real = lf_col(EC_IS_REAL), ets = lf_col(EC_TS);
    const orc_lf ec_lo = lf_col(EC_C_LO), ec_hi = lf_col(EC_C_HI), em_lo = lf_col(EC_M_LO), em_hi = lf_col(EC_M_HI);
    orc_inter* it = &g_ecall[0];
    memset(it, 0, sizeof *it);
    it->bus = BUS_ECALL; it->sign = -1; it->mult = real; it->n_el = 6;
    it->el[0] = ets; it->el[1] = lf_col(EC_PC); it->el[2] = lf_col(EC_NP); it->el[3] = lf_col(EC_B_LO);
    it->el[4] = lf_col(EC_A_LO); it->el[5] = lf_col(EC_A_HI);
    /* a0 is read at ts + 1, a1 at ts + 2: previous access times ts + q - 1 - difference */
    orc_lf epts[2];
    for (int q = 0; q < 2; ++q) {
      lf_zero(&epts[q]);
      lf_add(&epts[q], EC_TS, 1); lf_add(&epts[q], EC_GAP + 2 * q, FP - 1); lf_add(&epts[q], EC_GAP + 2 * q + 1, FP - 65536);
      epts[q].c0 = (uint32_t)q;
    }
    g_ecall[1] = mem_inter(-1, real, lf_const(10), ec_lo, ec_hi, epts[0]);
    g_ecall[2] = mem_inter(+1, real, lf_const(10), ec_lo, ec_hi, lf_plus(ets, 1));
    g_ecall[3] = mem_inter(-1, real, lf_const(11), em_lo, em_hi, epts[1]);
    g_ecall[4] = mem_inter(+1, real, lf_const(11), em_lo, em_hi, lf_plus(ets, 2));
    g_ecall[5] = range_inter(-1, real, lf_const(0), lf_col(EC_GAP));
    g_ecall[6] = range_inter(-1, real, lf_const(0), lf_col(EC_GAP + 2));
    g_ecall[7] = bytes_inter(-1, real, lf_col(EC_GAP + 1), lf_col(EC_GAP + 3));
    it = &g_ecall[8];
    memset(it, 0, sizeof *it);
    it->bus = BUS_PUBC; it->sign = +1; it->n_el = 4;
    it->mult = lf_pair(EC_SC + SC_COMMIT, EC_SC + SC_DEFER, 1);
    it->el[0] = lf_pair(EC_SC + SC_COMMIT, EC_SC + SC_DEFER, 2);
    it->el[1] = ec_lo; it->el[2] = em_lo; it->el[3] = em_hi;
    it = &g_ecall[9];
    memset(it, 0, sizeof *it);
    it->bus = BUS_PUBH; it->sign = +1; it->mult = lf_col(EC_SC + SC_HALT); it->n_el = 2;
    it->el[0] = ec_lo; it->el[1] = ec_hi;
    /* a HINT_READ announces (pointer, words) to the hint chip; the word count is a 16-bit value */
    it = &g_ecall[10];
    memset(it, 0, sizeof *it);
    it->bus = BUS_HINTR; it->sign = +1; it->mult = lf_col(EC_SC + SC_HINT_READ); it->n_el = 2;
    it->el[0] = lf_pair(EC_C_LO, EC_C_HI, 65536); it->el[1] = lf_col(EC_NW);
    g_ecall[11] = range_inter(-1, lf_col(EC_SC + SC_HINT_READ), lf_const(0), lf_col(EC_NW));
    /* hint chip: a read's first word takes the announcement; a word the run touches hands its initial value to the memory
     * boundary chip over the IMG bus */
    it = &g_hint[0];
    memset(it, 0, sizeof *it);
    it->bus = BUS_HINTR; it->sign = -1; it->mult = lf_col(HN_FIRST); it->n_el = 2;
    it->el[0] = lf_col(HN_ADDR); it->el[1] = lf_col(HN_CNT);
    it = &g_hint[1];
    memset(it, 0, sizeof *it);
    it->bus = BUS_IMG; it->sign = +1; it->mult = lf_col(HN_USED); it->n_el = 3;
    it->el[0] = lf_col(HN_ADDR); it->el[1] = lf_col(HN_LO); it->el[2] = lf_col(HN_HI);
  }
  /* ---- keccak chip: on an export row, word i of the input and of the output state, i = 0..49 ---- */
  for (int i = 0; i < 50; ++i) {
    orc_inter* it = &g_keccak[i];
    memset(it, 0, sizeof *it);
    const int lane = i >> 1, half = i & 1;
    const int out = lane == 0 ? KA_APPP00 + 2 * half : KA_APP + 4 * lane + 2 * half;
    it->bus = BUS_KIO; it->sign = +1; it->mult = lf_col(KA_EXPORT); it->n_el = 6;
    it->el[0] = lf_col(KC_TS); it->el[1] = lf_const((uint32_t)i);
    it->el[2] = lf_col(KA_PREIMAGE + 4 * lane + 2 * half); it->el[3] = lf_col(KA_PREIMAGE + 4 * lane + 2 * half + 1);
    it->el[4] = lf_col(out); it->el[5] = lf_col(out + 1);
  }
  /* ---- keccak-memory ---- */
  {
    orc_inter* it = &g_kmem[0];
    memset(it, 0, sizeof *it);
    it->bus = BUS_KCALL; it->sign = -1; it->mult = lf_col(KM_CALL); it->n_el = 3;
    it->el[0] = lf_col(KM_TS); it->el[1] = lf_col(KM_PTR_LO); it->el[2] = lf_col(KM_PTR_HI);
    it = &g_kmem[1];
    memset(it, 0, sizeof *it);
    it->bus = BUS_KIO; it->sign = -1; it->mult = lf_col(KM_IS_REAL); it->n_el = 6;
    it->el[0] = lf_col(KM_TS); it->el[1] = lf_col(KM_IDX); it->el[2] = lf_col(KM_OLD_LO); it->el[3] = lf_col(KM_OLD_HI);
    it->el[4] = lf_col(KM_NEW_LO); it->el[5] = lf_col(KM_NEW_HI);
    g_kmem[2] = mem_inter(-1, lf_col(KM_IS_REAL), lf_col(KM_ADDR), lf_col(KM_OLD_LO), lf_col(KM_OLD_HI), lf_col(KM_PTS));
    g_kmem[3] = mem_inter(+1, lf_col(KM_IS_REAL), lf_col(KM_ADDR), lf_col(KM_NEW_LO), lf_col(KM_NEW_HI),
                          lf_plus(lf_col(KM_TS), 2));
    g_kmem[4] = range_inter(-1, lf_col(KM_IS_REAL), zero, lf_col(KM_GL));
    g_kmem[5] = bytes_inter(-1, lf_col(KM_IS_REAL), lf_col(KM_GH), zero);
    /* the state pointer is word-aligned, lies at 0x10000 or above (not in register space) and the 200 bytes end below
     * 0x78000000: its high limb is looked up as kind 2 (1 .. ADDR_HI_MAX) */
    g_kmem[6] = range_inter(-1, lf_col(KM_CALL), one, lf_col(KM_PTR_LO));
    g_kmem[7] = range_inter(-1, lf_col(KM_CALL), lf_const(2), lf_col(KM_PTR_HI));
  }
  /* ---- memory boundary ---- */
  {
    const orc_lf real = lf_col(MF_IS_REAL), init = lf_col(MF_IS_INIT), addr = lf_pair(MF_LO, MF_HI, 65536);
    g_memfinal[0] = mem_inter(-1, real, addr, lf_col(MF_FIN_LO), lf_col(MF_FIN_HI), lf_col(MF_FIN_TS));
    g_memfinal[1] = mem_inter(+1, real, addr, lf_col(MF_INIT_LO), lf_col(MF_INIT_HI), zero);
    orc_inter* it = &g_memfinal[2];
    memset(it, 0, sizeof *it);
    it->bus = BUS_IMG; it->sign = -1; it->mult = lf_pair(MF_IS_REAL, MF_IS_ZERO, FP - 1); it->n_el = 3; /* an image word or a hinted one */
    it->el[0] = addr; it->el[1] = lf_col(MF_INIT_LO); it->el[2] = lf_col(MF_INIT_HI);
    g_memfinal[3] = range_inter(-1, real, zero, lf_col(MF_LO));
    g_memfinal[4] = range_inter(-1, real, zero, lf_col(MF_HI));
    g_memfinal[5] = range_inter(-1, real, zero, lf_const_minus(ADDR_HI_MAX, MF_HI));
    g_memfinal[6] = range_inter(-1, real, zero, lf_col(MF_D_LO));
    g_memfinal[7] = range_inter(-1, real, zero, lf_col(MF_D_HI));
    g_memfinal[8] = range_inter(-1, init, zero, lf_col(MF_INIT_LO));
    g_memfinal[9] = range_inter(-1, init, zero, lf_col(MF_INIT_HI));
  }
  /* ---- image / program / table: the row is [preprocessed | main] ---- */
  {
    orc_inter* it = &g_image[0];
    memset(it, 0, sizeof *it);
    it->bus = BUS_IMG; it->sign = +1; it->mult = lf_col(IMAGE_PREP_WIDTH + 0); it->n_el = 3;
    it->el[0] = lf_col(IMG_P_ADDR); it->el[1] = lf_col(IMG_P_LO); it->el[2] = lf_col(IMG_P_HI);
    it = &g_program[0];
    memset(it, 0, sizeof *it);
    it->bus = BUS_PROG; it->sign = +1; it->mult = lf_col(PROGRAM_PREP_WIDTH + 0); it->n_el = 13;
    for (int j = 0; j < 13; ++j) it->el[j] = lf_col(j);
    const orc_lf idx = lf_pair(TB_P_X, TB_P_Y, 256);
    g_table[0] = range_inter(+1, lf_col(TABLE_PREP_WIDTH + TB_M_R16), zero, idx);
    g_table[1] = range_inter(+1, lf_col(TABLE_PREP_WIDTH + TB_M_AL), one, idx);
    g_table[2] = range_inter(+1, lf_col(TABLE_PREP_WIDTH + TB_M_TOP), lf_const(2), idx);
    g_table[3] = bytes_inter(+1, lf_col(TABLE_PREP_WIDTH + TB_M_BY), lf_col(TB_P_X), lf_col(TB_P_Y));
    /* byte operations (kind, x, y, z): 1 xor, 2 or (= x + y - and), 3 and */
    for (int k = 0; k < 3; ++k) {
      it = &g_table[4 + k];
      memset(it, 0, sizeof *it);
      it->bus = BUS_BYTEOP; it->sign = +1; it->mult = lf_col(TABLE_PREP_WIDTH + TB_M_XOR + k); it->n_el = 4;
      it->el[0] = lf_const((uint32_t)k + 1); it->el[1] = lf_col(TB_P_X); it->el[2] = lf_col(TB_P_Y);
      if (k == 0) it->el[3] = lf_col(TB_P_XOR);
      else if (k == 2) it->el[3] = lf_col(TB_P_AND);
      else { it->el[3] = lf_pair(TB_P_X, TB_P_Y, 1); lf_add(&it->el[3], TB_P_AND, FP - 1); }
    }
  }
  /* ---- multiplier ---- */
  for (int hi = 0; hi < 2; ++hi) {
    orc_inter* it = &g_mul[hi];
    memset(it, 0, sizeof *it);
    it->bus = BUS_ALU; it->sign = -1; it->n_el = 7;
    if (hi) it->mult = lf_col(MU_HI);
    else { it->mult = lf_pair(MU_IS_REAL, MU_HI, FP - 1); lf_add(&it->mult, MU_SH, FP - 1); lf_add(&it->mult, MU_SHU, FP - 1); }
    it->el[0] = lf_const(hi ? OP_MULHU : OP_MUL);
    it->el[1] = lf_bits(MU_P + 32 * hi, 16); it->el[2] = lf_bits(MU_P + 32 * hi + 16, 16);
    it->el[3] = lf_bits(MU_B, 16); it->el[4] = lf_bits(MU_B + 16, 16); it->el[5] = lf_bits(MU_C, 16); it->el[6] = lf_bits(MU_C + 16, 16);
  }
  {
    /* mulh / mulhsu: the signed high word R, range-checked */
    orc_inter* it = &g_mul[2];
    memset(it, 0, sizeof *it);
    it->bus = BUS_ALU; it->sign = -1; it->n_el = 7;
    it->mult = lf_pair(MU_SH, MU_SHU, 1);
    lf_zero(&it->el[0]); lf_add(&it->el[0], MU_SH, OP_MULH); lf_add(&it->el[0], MU_SHU, OP_MULHSU);
    it->el[1] = lf_col(MU_R); it->el[2] = lf_col(MU_R + 1);
    it->el[3] = lf_bits(MU_B, 16); it->el[4] = lf_bits(MU_B + 16, 16); it->el[5] = lf_bits(MU_C, 16); it->el[6] = lf_bits(MU_C + 16, 16);
    g_mul[3] = range_inter(-1, lf_pair(MU_SH, MU_SHU, 1), zero, lf_col(MU_R));
    g_mul[4] = range_inter(-1, lf_pair(MU_SH, MU_SHU, 1), zero, lf_col(MU_R + 1));
  }
  /* ---- divider: the instruction from the CPU row, the product |q| |d| from the multiplier chip (low word PL, high word
   * zero), range lookups ---- */
  {
    const orc_lf real = lf_col(DV_IS_REAL), nzd = lf_col(DV_NZD), sgn = lf_pair(DV_F + 0, DV_F + 2, 1);
    orc_inter* it = &g_div[0];
    memset(it, 0, sizeof *it);
    it->bus = BUS_ALU; it->sign = -1; it->mult = real; it->n_el = 7;
    lf_zero(&it->el[0]);
    lf_add(&it->el[0], DV_F + 0, OP_DIV); lf_add(&it->el[0], DV_F + 1, OP_DIVU); lf_add(&it->el[0], DV_F + 2, OP_REM); lf_add(&it->el[0], DV_F + 3, OP_REMU);
    it->el[1] = lf_col(DV_A); it->el[2] = lf_col(DV_A + 1); it->el[3] = lf_col(DV_N); it->el[4] = lf_col(DV_N + 1);
    it->el[5] = lf_col(DV_D); it->el[6] = lf_col(DV_D + 1);
    for (int hi = 0; hi < 2; ++hi) {
      it = &g_div[1 + hi];
      memset(it, 0, sizeof *it);
      it->bus = BUS_ALU; it->sign = +1; it->mult = nzd; it->n_el = 7;
      it->el[0] = lf_const(hi ? OP_MULHU : OP_MUL);
      it->el[1] = hi ? zero : lf_col(DV_PL); it->el[2] = hi ? zero : lf_col(DV_PL + 1);
      it->el[3] = lf_col(DV_AQ); it->el[4] = lf_col(DV_AQ + 1); it->el[5] = lf_col(DV_AD); it->el[6] = lf_col(DV_AD + 1);
    }
    static const int checked[8] = {DV_A, DV_A + 1, DV_AN, DV_AN + 1, DV_AR, DV_AR + 1, DV_E, DV_E + 1};
    for (int k = 0; k < 8; ++k) g_div[3 + k] = range_inter(-1, real, zero, lf_col(checked[k]));
    orc_lf nh2, dh2;
    lf_zero(&nh2); lf_add(&nh2, DV_NH, 2);
    lf_zero(&dh2); lf_add(&dh2, DV_DH, 2);
    g_div[11] = range_inter(-1, sgn, zero, nh2);
    g_div[12] = range_inter(-1, sgn, zero, dh2);
  }
  /* ---- ALU ---- */
  {
    orc_inter* it = &g_alu[0];
    memset(it, 0, sizeof *it);
    it->bus = BUS_ALU; it->sign = -1; it->mult = lf_col(AL_IS_REAL); it->n_el = 7;
    lf_zero(&it->el[0]);
    for (int k = 0; k < 4; ++k) lf_add(&it->el[0], AL_SEL + k, (uint32_t)(OP_SLL + k));
    it->el[1] = lf_col(AL_A); it->el[2] = lf_col(AL_A + 1);
    it->el[3] = lf_bits(AL_B, 16); it->el[4] = lf_bits(AL_B + 16, 16); it->el[5] = lf_bits(AL_C, 16); it->el[6] = lf_bits(AL_C + 16, 16);
  }
  /* ---- bitwise: the word tuple from the CPU row, and one byte-operation lookup per byte ---- */
  {
    orc_inter* it = &g_bw[0];
    memset(it, 0, sizeof *it);
    it->bus = BUS_ALU; it->sign = -1; it->mult = lf_col(BW_IS_REAL); it->n_el = 7;
    lf_zero(&it->el[0]);
    for (int k = 0; k < 3; ++k) lf_add(&it->el[0], BW_SEL + k, (uint32_t)(OP_XOR + k));
    for (int w = 0; w < 3; ++w) { /* a, b, c as limbs of two bytes */
      const int base = w == 0 ? BW_A : w == 1 ? BW_B : BW_C;
      it->el[1 + 2 * w] = lf_pair(base, base + 1, 256);
      it->el[2 + 2 * w] = lf_pair(base + 2, base + 3, 256);
    }
    for (int i = 0; i < 4; ++i) {
      it = &g_bw[1 + i];
      memset(it, 0, sizeof *it);
      it->bus = BUS_BYTEOP; it->sign = -1; it->mult = lf_col(BW_IS_REAL); it->n_el = 4;
      lf_zero(&it->el[0]);
      for (int k = 0; k < 3; ++k) lf_add(&it->el[0], BW_SEL + k, (uint32_t)k + 1);
      it->el[1] = lf_col(BW_B + i); it->el[2] = lf_col(BW_C + i); it->el[3] = lf_col(BW_A + i);
    }
  }
  /* ---- Poseidon2 (machine.h "Poseidon2 chip").  DIGEST tuples are (tag, type, key, mask, 8 words); type 2: a heap node
   * (stage 1: children in, parent out), type 1: the hash of an injected matrix row (a sponge's last row -> the injection
   * row with the same labels), type 0: the end of a run (-> the verifier, who knows the root).  The output words are the
   * external linear layer applied to the last round's S-box outputs (circ(2 M4, M4, M4, M4),
   * M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]) ---- */
  {
    static const uint32_t m4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    const int ylast = P2_EXT + 32 * 7 + 16;
    const orc_lf tag = lf_col(P2_T), mask = lf_col(P2_M), key = lf_pair(P2_KL, P2_KH, 65536);
    for (int side = 0; side < 2; ++side) {
      orc_inter* it = &g_p2[side];
      memset(it, 0, sizeof *it);
      it->bus = BUS_DIGEST; it->sign = -1; it->mult = lf_col(P2_FN); it->n_el = 12;
      it->el[0] = tag; it->el[1] = lf_const(2);
      lf_zero(&it->el[2]); lf_add(&it->el[2], P2_KL, 2); lf_add(&it->el[2], P2_KH, 2 * 65536); it->el[2].c0 = (uint32_t)side;
      it->el[3] = mask;
      for (int j = 0; j < 8; ++j) it->el[4 + j] = lf_col(P2_IN + 8 * side + j);
    }
    orc_inter* it = &g_p2[2];
    memset(it, 0, sizeof *it);
    it->bus = BUS_DIGEST; it->sign = -1; it->mult = lf_col(P2_FJ); it->n_el = 12;
    it->el[0] = tag; it->el[1] = lf_const(1); it->el[2] = key; it->el[3] = mask;
    for (int j = 0; j < 8; ++j) it->el[4 + j] = lf_col(P2_IN + 8 + j);
    it = &g_p2[3];
    memset(it, 0, sizeof *it);
    it->bus = BUS_DIGEST; it->sign = +1; it->mult = lf_pair(P2_FN, P2_SND, 1); it->n_el = 12;
    it->el[0] = tag;
    lf_zero(&it->el[1]); lf_add(&it->el[1], P2_FN, 2); lf_add(&it->el[1], P2_SZ, 1); lf_add(&it->el[1], P2_SC, 1);
    it->el[2] = key; it->el[3] = mask;
    for (int j = 0; j < 8; ++j) {
      lf_zero(&it->el[4 + j]);
      for (int i = 0; i < 16; ++i) lf_add(&it->el[4 + j], ylast + i, m4[j & 3][i & 3] * ((i >> 2) == (j >> 2) ? 2u : 1u));
    }
    /* the pair a FRI leaf hashes goes to the fold chip */
    it = &g_p2[4];
    memset(it, 0, sizeof *it);
    it->bus = BUS_PAIR; it->sign = +1; it->mult = lf_col(P2_FR); it->n_el = 9;
    it->el[0] = tag;
    for (int j = 0; j < 8; ++j) it->el[1 + j] = lf_col(P2_IN + j);
    /* the key's limbs: 16 bits, and twice the high limb plus one in 1 .. ADDR_HI_MAX (kind 2): the key stays below 0x3C000000 and
     * its children's keys 2K, 2K + 1 below 0x78000000 < p - no key aliases another mod p */
    g_p2[5] = range_inter(-1, lf_col(P2_IS_REAL), lf_const(0), lf_col(P2_KL));
    { orc_lf kh2; lf_zero(&kh2); lf_add(&kh2, P2_KH, 2); kh2.c0 = 1; g_p2[6] = range_inter(-1, lf_col(P2_IS_REAL), lf_const(2), kh2); }
  }
  /* ---- stage 2b: the Poseidon2 chip's run ends and hash ends ---- */
  {
    static const uint32_t m4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    const int ylast = P2_EXT + 32 * 7 + 16;
    const orc_lf tag = lf_col(P2_T), mask = lf_col(P2_M), key = lf_pair(P2_KL, P2_KH, 65536);
    orc_inter* it = &g_p2[7];
    memset(it, 0, sizeof *it);
    it->bus = BUS_POS; it->sign = +1; it->mult = lf_col(P2_RE); it->n_el = 4;
    it->el[0] = tag; it->el[1] = key; it->el[2] = mask; it->el[3] = lf_col(P2_RID);
    it = &g_p2[8];
    memset(it, 0, sizeof *it);
    it->bus = BUS_ROOT; it->sign = -1; it->mult = lf_col(P2_RE); it->n_el = 9;
    it->el[0] = lf_col(P2_RID);
    for (int j = 0; j < 8; ++j) {
      lf_zero(&it->el[1 + j]);
      for (int i = 0; i < 16; ++i) lf_add(&it->el[1 + j], ylast + i, m4[j & 3][i & 3] * ((i >> 2) == (j >> 2) ? 2u : 1u));
    }
    it = &g_p2[9];
    memset(it, 0, sizeof *it);
    it->bus = BUS_SEG; it->sign = +1; it->mult = lf_col(P2_SE); it->n_el = 11;
    it->el[0] = tag; it->el[1] = key; it->el[2] = mask;
    for (int j = 0; j < 4; ++j) { it->el[3 + j] = lf_col(P2_SO + j); it->el[7 + j] = lf_col(P2_AP + j); }
  }
  /* ---- query chip (machine.h).  T(r) = 1 + 64 QL + 2^18 LEAF + r, RID(r) = 64 LEAF + r ---- */
  {
#define QTAG(f, r, kcol) do { lf_zero(&(f)); lf_add(&(f), QR_QL, LEAF_TAG_STRIDE); lf_add(&(f), QR_LEAF, LEAF_TAG_LEAF_STRIDE); \
                              if ((kcol) >= 0) lf_add(&(f), (kcol), 1); (f).c0 = 1 + (r); } while (0)
#define QRID(f, r, kcol) do { lf_zero(&(f)); lf_add(&(f), QR_LEAF, 64); if ((kcol) >= 0) lf_add(&(f), (kcol), 1); (f).c0 = (r); } while (0)
    const orc_lf leaf = lf_col(QR_LEAF), last = lf_col(QR_LAST), lay = lf_col(QR_LAY), hasro = lf_col(QR_HASRO);
    int n = 0;
    orc_inter* it = &g_qr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_QIDX; it->sign = -1; it->mult = last; it->n_el = 3;
    it->el[0] = leaf; it->el[1] = lf_col(QR_QL); it->el[2] = lf_col(QR_ACC);
    it = &g_qr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_LEAFK; it->sign = -1; it->mult = last; it->n_el = 3;
    it->el[0] = leaf; it->el[1] = lf_col(QR_K); it->el[2] = lf_col(QR_OMI);
    it = &g_qr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_FINAL; it->sign = -1; it->mult = last; it->n_el = 5;
    it->el[0] = leaf;
    for (int j = 0; j < 4; ++j) it->el[1 + j] = lf_col(QR_F + j);
    for (uint32_t r = 1; r <= 3; ++r) {
      it = &g_qr[n++];
      memset(it, 0, sizeof *it);
      it->bus = BUS_POS; it->sign = -1; it->mult = lf_col(QR_CSR); it->n_el = 4;
      QTAG(it->el[0], r, -1);
      lf_zero(&it->el[1]); lf_add(&it->el[1], QR_POW, 2); lf_add(&it->el[1], QR_LOW, 2); lf_add(&it->el[1], QR_BIT, 1);
      it->el[2] = lf_col(QR_MT); QRID(it->el[3], r, -1);
    }
    it = &g_qr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_POS; it->sign = -1; it->mult = lf_col(QR_PR0); it->n_el = 4;
    QTAG(it->el[0], 0, -1);
    lf_zero(&it->el[1]); lf_add(&it->el[1], QR_POW, 2); lf_add(&it->el[1], QR_LOW, 2); lf_add(&it->el[1], QR_CS, 1);
    it->el[2] = lf_col(QR_MT0); QRID(it->el[3], 0, -1);
    it = &g_qr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_POS; it->sign = -1; it->mult = lay; it->n_el = 4;
    QTAG(it->el[0], 4, QR_K);
    lf_zero(&it->el[1]); lf_add(&it->el[1], QR_POW, 2); lf_add(&it->el[1], QR_REV, 2); lf_add(&it->el[1], QR_CS, 1);
    it->el[2] = lf_const(0); QRID(it->el[3], 4, QR_K);
    it = &g_qr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_PAIR; it->sign = -1; it->mult = lay; it->n_el = 9;
    QTAG(it->el[0], 4, QR_K);
    for (int j = 0; j < 4; ++j) { it->el[1 + j] = lf_col(QR_LO + j); it->el[5 + j] = lf_col(QR_HI + j); }
    it = &g_qr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_BETA; it->sign = -1; it->mult = lay; it->n_el = 6;
    it->el[0] = leaf; it->el[1] = lf_col(QR_K);
    for (int j = 0; j < 4; ++j) it->el[2 + j] = lf_col(QR_BETA + j);
    for (uint32_t r = 0; r <= 3; ++r) {
      it = &g_qr[n++];
      memset(it, 0, sizeof *it);
      it->bus = BUS_SEG; it->sign = -1; it->mult = r == 0 ? lf_col(QR_HAS0) : hasro; it->n_el = 11;
      QTAG(it->el[0], r, -1);
      it->el[1] = lf_col(r == 0 ? QR_KEY0 : QR_KEYJ); it->el[2] = lf_col(r == 0 ? QR_M0 : QR_MJ);
      for (int j = 0; j < 4; ++j) { it->el[3 + j] = lf_col(QR_H + 4 * (int)r + j); it->el[7 + j] = lf_col(QR_AF + j); }
    }
    it = &g_qr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_ZETA; it->sign = -1; it->mult = hasro; it->n_el = 5;
    it->el[0] = leaf;
    for (int j = 0; j < 4; ++j) it->el[1 + j] = lf_col(QR_ZETA + j);
    it = &g_qr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_AF; it->sign = -1; it->mult = hasro; it->n_el = 9;
    it->el[0] = leaf;
    for (int j = 0; j < 4; ++j) { it->el[1 + j] = lf_col(QR_AF + j); it->el[5 + j] = lf_col(QR_DL + j); }
    it = &g_qr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_BCONST; it->sign = -1; it->mult = hasro; it->n_el = 12;
    it->el[0] = leaf; it->el[1] = lf_col(QR_K); it->el[2] = lf_col(QR_HAS0); it->el[3] = lf_col(QR_WH);
    for (int j = 0; j < 4; ++j) { it->el[4 + j] = lf_col(QR_B1 + j); it->el[8 + j] = lf_col(QR_B2 + j); }
    if (n != 17) abort();
#undef QTAG
#undef QRID
  }
  /* ---- transcript chip (machine.h).  The output words of a row: the external linear layer of the last round's columns ---- */
  {
    static const uint32_t m4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    const int ylast = TR_EXT + 32 * 7 + 16;
    orc_lf ow[8];
    for (int j = 0; j < 8; ++j) {
      lf_zero(&ow[j]);
      for (int i = 0; i < 16; ++i) lf_add(&ow[j], ylast + i, m4[j & 3][i & 3] * ((i >> 2) == (j >> 2) ? 2u : 1u));
    }
    const orc_lf leaf = lf_col(TR_LEAF), step = lf_col(TR_STEP);
    orc_lf flags;
    lf_zero(&flags);
    for (int k = 0; k < 7; ++k) lf_add(&flags, TR_UROOT + k, 1u << k);
    for (int k = 0; k < 8; ++k) lf_add(&flags, TR_QM + k, 128u << k);
    int n = 0;
    orc_inter* it = &g_tr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_TBLK; it->sign = -1; it->mult = lf_col(TR_ABS); it->n_el = 12;
    it->el[0] = leaf; it->el[1] = step; it->el[2] = flags; it->el[3] = lf_col(TR_RIDK);
    for (int j = 0; j < 8; ++j) it->el[4 + j] = lf_col(TR_IN + j);
    it = &g_tr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_TSQ; it->sign = -1; it->mult = lf_pair(TR_IS_REAL, TR_ABS, FP - 1); it->n_el = 4;
    it->el[0] = leaf; it->el[1] = step; it->el[2] = flags; it->el[3] = lf_col(TR_QBASE);
    it = &g_tr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_ROOT; it->sign = +1; it->mult = lf_col(TR_MROOT); it->n_el = 9;
    lf_zero(&it->el[0]); lf_add(&it->el[0], TR_LEAF, 64); lf_add(&it->el[0], TR_RIDK, 1);
    for (int j = 0; j < 8; ++j) it->el[1 + j] = lf_col(TR_IN + j);
    it = &g_tr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_FINAL; it->sign = +1; it->mult = lf_col(TR_MFIN); it->n_el = 5;
    it->el[0] = leaf;
    for (int j = 0; j < 4; ++j) it->el[1 + j] = lf_col(TR_IN + j);
    it = &g_tr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_ZETA; it->sign = +1; it->mult = lf_col(TR_MZETA); it->n_el = 5;
    it->el[0] = leaf;
    for (int j = 0; j < 4; ++j) it->el[1 + j] = ow[7 - j];
    it = &g_tr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_AF; it->sign = +1; it->mult = lf_col(TR_MAF); it->n_el = 9;
    it->el[0] = leaf;
    for (int j = 0; j < 8; ++j) it->el[1 + j] = ow[7 - j];
    it = &g_tr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_BETA; it->sign = +1; it->mult = lf_col(TR_MBETA); it->n_el = 6;
    it->el[0] = leaf; it->el[1] = lf_plus(lf_col(TR_RIDK), FP - 4);
    for (int j = 0; j < 4; ++j) it->el[2 + j] = ow[7 - j];
    it = &g_tr[n++];
    memset(it, 0, sizeof *it);
    it->bus = BUS_POW; it->sign = +1; it->mult = lf_col(TR_UPOW); it->n_el = 2;
    it->el[0] = leaf; it->el[1] = ow[7];
    for (int j = 0; j < 8; ++j) {
      it = &g_tr[n++];
      memset(it, 0, sizeof *it);
      it->bus = BUS_QIDX; it->sign = +1; it->mult = lf_col(TR_QM + j); it->n_el = 3;
      it->el[0] = leaf; it->el[1] = lf_plus(lf_col(TR_QBASE), (uint32_t)j); it->el[2] = ow[7 - j];
    }
    if (n != 16) abort();
  }
  /* ---- sub-word ---- */
  {
    static const uint32_t codes[6] = {OP_LB, OP_LH, OP_LBU, OP_LHU, OP_SB, OP_SH};
    orc_inter* it = &g_sub[0];
    memset(it, 0, sizeof *it);
    it->bus = BUS_SUB; it->sign = -1; it->mult = lf_col(SW_IS_REAL); it->n_el = 7;
    lf_zero(&it->el[0]);
    for (int k = 0; k < 6; ++k) lf_add(&it->el[0], SW_SEL + k, codes[k]);
    lf_zero(&it->el[1]); lf_add(&it->el[1], SW_O + 1, 1); lf_add(&it->el[1], SW_O + 2, 2); lf_add(&it->el[1], SW_O + 3, 3);
    it->el[2] = lf_col(SW_A); it->el[3] = lf_col(SW_A + 1);
    it->el[4] = lf_pair(SW_MB, SW_MB + 1, 256); it->el[5] = lf_pair(SW_MB + 2, SW_MB + 3, 256); it->el[6] = lf_pair(SW_CB, SW_CB + 1, 256);
    /* the bytes are bytes; the sign bit of a signed load is bit 7 of the byte it extends */
    g_sub[1] = bytes_inter(-1, lf_col(SW_IS_REAL), lf_col(SW_MB), lf_col(SW_MB + 1));
    g_sub[2] = bytes_inter(-1, lf_col(SW_IS_REAL), lf_col(SW_MB + 2), lf_col(SW_MB + 3));
    g_sub[3] = bytes_inter(-1, lf_col(SW_IS_REAL), lf_col(SW_CB), lf_col(SW_CB + 1));
    it = &g_sub[4];
    memset(it, 0, sizeof *it);
    it->bus = BUS_BYTEOP; it->sign = -1; it->mult = lf_pair(SW_SEL + 0, SW_SEL + 1, 1); it->n_el = 4;
    it->el[0] = lf_const(3); /* and */
    it->el[1] = lf_col(SW_SELB); it->el[2] = lf_const(0x80);
    lf_zero(&it->el[3]); lf_add(&it->el[3], SW_S, 128);
  }
  g_chips[CH_TABLE] = (orc_chip){"table", TABLE_PREP_WIDTH, TABLE_WIDTH, 7, g_table, 0, 0};
  g_chips[CH_CPU] = (orc_chip){"cpu", 0, CPU_WIDTH, CPU_INTER, g_cpu, 0, 5};   /* ALU, SUB (loads), SUB (stores), KCALL, ECALL: one class each */
  g_chips[CH_CPU2] = (orc_chip){"cpu2", 0, CPU_WIDTH, CPU_INTER, g_cpu, 0, 5};
  {
    static const char* names[CPU_INST] = {"cpu", "cpu2", "cpu3", "cpu4", "cpu5", "cpu6", "cpu7", "cpu8"};
    for (int i = 2; i < CPU_INST; ++i) g_chips[orc_cpu_chip(i)] = (orc_chip){names[i], 0, CPU_WIDTH, CPU_INTER, g_cpu, 0, 5};
  }
  g_chips[CH_KECCAK] = (orc_chip){"keccak", 0, KECCAK_WIDTH, 50, g_keccak, 0, 0};
  g_chips[CH_KMEM] = (orc_chip){"keccak-mem", 0, KMEM_WIDTH, 8, g_kmem, 0, 0};
  g_chips[CH_MEMFINAL] = (orc_chip){"mem-final", 0, MEMFINAL_WIDTH, 10, g_memfinal, 0, 0};
  g_chips[CH_IMAGE] = (orc_chip){"image", IMAGE_PREP_WIDTH, IMAGE_WIDTH, 1, g_image, 0, 0};
  g_chips[CH_PROGRAM] = (orc_chip){"program", PROGRAM_PREP_WIDTH, PROGRAM_WIDTH, 1, g_program, 0, 0};
  g_chips[CH_MUL] = (orc_chip){"mul", 0, MUL_WIDTH, 5, g_mul, 0, 0};
  g_chips[CH_DIV] = (orc_chip){"divider", 0, DIV_WIDTH, 13, g_div, 0, 0};
  g_chips[CH_ALU] = (orc_chip){"alu", 0, ALU_WIDTH, 1, g_alu, 0, 0};
  g_chips[CH_ALU2] = (orc_chip){"alu2", 0, ALU_WIDTH, 1, g_alu, 0, 0};
  g_chips[CH_SUB] = (orc_chip){"subword", 0, SUB_WIDTH, 5, g_sub, 0, 0};
  g_chips[CH_SUB2] = (orc_chip){"subword2", 0, SUB_WIDTH, 5, g_sub, 0, 0};
  g_chips[CH_BW] = (orc_chip){"bitwise", 0, BW_WIDTH, 5, g_bw, 0, 0};
  g_chips[CH_BW2] = (orc_chip){"bitwise2", 0, BW_WIDTH, 5, g_bw, 0, 0};
  g_chips[CH_P2] = (orc_chip){"poseidon2", 0, P2CHIP_WIDTH, 10, g_p2, 0, 0};
  g_chips[CH_QR] = (orc_chip){"query", 0, QR_WIDTH, 17, g_qr, 0, 0};
  g_chips[CH_TR] = (orc_chip){"transcript", 0, TR_WIDTH, 16, g_tr, 0, 0};
  g_chips[CH_ECALL] = (orc_chip){"ecall", 0, ECALL_WIDTH, 12, g_ecall, 0, 0};
  g_chips[CH_HINT] = (orc_chip){"hint", 0, HINT_WIDTH, 2, g_hint, 0, 0};
  g_ready = 1;
  for (int c = 0; c < N_CHIPS; ++c) g_chips[c].n_constraints = count_constraints(c);
}

const orc_chip* orc_machine_chip(int chip) {
  build();
  return &g_chips[chip];
}

/* ------------------------------------------------------------------------------------------
 * heights, event lists
 * ---------------------------------------------------------------------------------------- */
static int clog2(size_t v) {
  int l = 0;
  while (((size_t)1 << l) < v) ++l;
  return l;
}
static int at_least5(int l) { return l < 5 ? 5 : l; }

/* rows of the first of two instances: the largest power of two strictly below the count (at least 32) */
static size_t split_rows(size_t n) {
  size_t h0 = 32;
  while (2 * h0 < n) h0 *= 2;
  return h0;
}
static int second_logh(size_t n) {
  const size_t h0 = split_rows(n);
  return at_least5(clog2(n > h0 ? n - h0 : 1));
}
static const uint32_t* prog_row(const orc_machine_input* in, uint32_t pc) { return in->program + 9 * (size_t)((pc - in->text_base) >> 2); }

size_t orc_machine_events(const orc_machine_input* in, int which, uint32_t* out) {
  size_t n = 0;
  for (size_t g = 0; g < in->n_cycles; ++g)
    if (event_kind(prog_row(in, in->cycles[12 * g])[1]) == which) {
      if (out) out[n] = (uint32_t)g;
      ++n;
    }
  return n;
}
uint32_t orc_machine_x0_last(const orc_machine_input* in) {
  for (size_t g = in->n_cycles; g-- > 0;) {
    const uint32_t* p = prog_row(in, in->cycles[12 * g]);
    const uint32_t ts = 4 * ((uint32_t)g + 1);
    if (p[3] && p[6] == 0) return ts + 1; /* rs2 = x0 is read after rs1 */
    if (p[5] == 0) return ts;
  }
  return 0;
}
static uint32_t pad_pc_of(const orc_machine_input* in) { return in->text_base + 4 * (uint32_t)(in->n_program - 1); }

/* rows of the first instance of a split chip */
static size_t first_rows(const orc_machine_input* in, int chip, size_t n) {
  return in->shape ? (size_t)1 << in->shape[chip] : split_rows(n);
}
/* rows of the CPU instances together; first cycle (row0) of instance i */
static size_t cpu_row0(const int logh[N_CHIPS], int inst) {
  size_t r = 0;
  for (int i = 0; i < inst; ++i) r += (size_t)1 << logh[orc_cpu_chip(i)];
  return r;
}
static size_t cpu_rows(const orc_machine_input* in) {
  int logh[N_CHIPS];
  orc_machine_heights(in, logh);
  return cpu_row0(logh, CPU_INST);
}
/* public scalars of CPU instance `chip`: where it starts (pc, time), whether another instance follows and at which pc
 * (the hand-over pcs are header words) */
void orc_machine_cpu_pub(const orc_machine_input* in, int chip, uint32_t pub[CPUPUB_N]) {
  int logh[N_CHIPS];
  orc_machine_heights(in, logh);
  const int inst = orc_cpu_instance(chip);
  const size_t r0 = cpu_row0(logh, inst), r1 = cpu_row0(logh, inst + 1);
  const uint32_t pad = pad_pc_of(in);
  pub[CPUPUB_PAD_PC] = pad;
  pub[CPUPUB_START_PC] = inst == 0 ? in->entry : r0 < in->n_cycles ? in->cycles[12 * r0] : pad;
  pub[CPUPUB_START_TS] = 4 * ((uint32_t)r0 + 1);
  pub[CPUPUB_HAS_SUCC] = inst + 1 < CPU_INST;
  pub[CPUPUB_END_PC] = inst + 1 < CPU_INST ? (r1 < in->n_cycles ? in->cycles[12 * r1] : pad) : 0;
}

/* words covered by the run's HINT_READs (ecalls with t0 = 0xf1: pointer in a0, length in a1), in execution order */
static size_t hint_words(const orc_machine_input* in) {
  const size_t n = orc_machine_events(in, 3, NULL);
  uint32_t* ev = (uint32_t*)malloc((n ? n : 1) * 4);
  orc_machine_events(in, 3, ev);
  size_t words = 0;
  for (size_t e = 0; e < n; ++e) {
    const uint32_t* cy = in->cycles + 12 * (size_t)ev[e];
    if (cy[2] == 0xf1) words += (cy[4] + 3) / 4;
  }
  free(ev);
  return words;
}

void orc_machine_heights(const orc_machine_input* in, int logh[N_CHIPS]) {
  if (in->shape) { memcpy(logh, in->shape, N_CHIPS * sizeof(int)); return; }
  const size_t na = orc_machine_events(in, 0, NULL), ns = orc_machine_events(in, 1, NULL), nb = orc_machine_events(in, 2, NULL);
  logh[CH_BW] = clog2(split_rows(nb));
  logh[CH_BW2] = second_logh(nb);
  {
    const int hc = at_least5(clog2((in->n_cycles + CPU_INST - 1) / CPU_INST));
    for (int i = 0; i < CPU_INST; ++i) logh[orc_cpu_chip(i)] = ((size_t)i << hc) < in->n_cycles || i == 0 ? hc : 5;
  }
  logh[CH_ALU] = clog2(split_rows(na));
  logh[CH_ALU2] = second_logh(na);
  logh[CH_SUB] = clog2(split_rows(ns));
  logh[CH_SUB2] = second_logh(ns);
  logh[CH_KECCAK] = at_least5(clog2(24 * in->n_keccak));
  logh[CH_KMEM] = at_least5(clog2(50 * in->n_keccak));
  logh[CH_MEMFINAL] = at_least5(clog2(in->n_memfinal));
  logh[CH_IMAGE] = in->log_image;
  logh[CH_PROGRAM] = in->log_prog;
  logh[CH_MUL] = at_least5(clog2(in->n_muls));
  logh[CH_TABLE] = TABLE_LOG_H;
  {
    size_t rows = orc_machine_agg_rows(in->agg_keys, in->agg_leaves, in->n_agg, NULL); /* one per ancestor of a supplied key */
    if (rows == (size_t)-1) rows = 0;
    rows += in->n_leaf_p2; /* ... and one per permutation of a leaf-proof check */
    logh[CH_P2] = at_least5(clog2(rows == 0 ? 1 : rows));
    logh[CH_QR] = at_least5(clog2(in->n_leaf_qr ? in->n_leaf_qr : 1));
    logh[CH_TR] = at_least5(clog2(in->n_leaf_tr ? in->n_leaf_tr : 1));
    logh[CH_HINT] = at_least5(clog2(hint_words(in) ? hint_words(in) : 1));
  }
  logh[CH_ECALL] = at_least5(clog2(orc_machine_events(in, 3, NULL)));
  logh[CH_DIV] = at_least5(clog2(orc_machine_events(in, 4, NULL)));
}

/* ------------------------------------------------------------------------------------------
 * trace generation
 * ---------------------------------------------------------------------------------------- */
static void put_bits(uint32_t* t, size_t h, size_t r, int col, uint32_t v, int n) {
  for (int i = 0; i < n; ++i) t[(size_t)(col + i) * h + r] = (v >> i) & 1u;
}
static void put_limbs(uint32_t* t, size_t h, size_t r, int col, uint32_t v) {
  t[(size_t)col * h + r] = v & 0xffff;
  t[(size_t)(col + 1) * h + r] = v >> 16;
}

typedef struct { uint32_t ts, ptr; uint64_t in[25]; uint32_t pts[50]; } kcall_t;

/* the flag a branch / set-less-than leaves: signed or unsigned b < c */
static uint32_t less_than(uint32_t code, uint32_t b, uint32_t c) { return code == OP_SLT ? ((int32_t)b < (int32_t)c) : (b < c); }

/* rows [0, h) of the instance whose first row is cycle `row0`; rows past the last cycle run the padding instruction */
static void fill_cpu(const orc_machine_input* in, size_t h, uint32_t* t, size_t row0) {
  const uint32_t pad_pc = pad_pc_of(in), x0_last = orc_machine_x0_last(in);
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < h; ++r) {
#define T(col) t[(size_t)(col) * h + r]
    const size_t g = row0 + r;  /* cycle index */
    const uint32_t ts = 4 * ((uint32_t)g + 1);
    T(C_TS) = ts;
    uint32_t gap[3] = {0, 0, 0};
    if (g >= in->n_cycles) {
      /* jal x0, 0 at the padding pc: reads x0 (rs1 = 0), writes nothing, jumps to itself */
      const uint32_t pts = g == in->n_cycles ? x0_last : ts - 4;
      T(C_PC) = pad_pc; T(C_NEXT_PC) = pad_pc; T(SELC(CL_JAL)) = 1;
      put_limbs(t, h, r, C_TGT_LO, pad_pc);
      gap[0] = ts - pts - 1;
    } else {
      const uint32_t* cy = in->cycles + 12 * g;
      const uint32_t pc = cy[0], b = cy[2], wprev = cy[6];
      uint32_t a = cy[1], c = cy[3], m = cy[4], mv = cy[5];
      const uint32_t* p = prog_row(in, pc);
      const uint32_t op = p[1], wr = p[2], rd = p[4], rs1 = p[5], imm = p[7], tgt = p[8];
      uint32_t use2 = p[3], rs2 = p[6];
      const int cls = orc_class_of(op);
      if (cls == CL_ECALL) { use2 = 0; rs2 = 0; c = 0; m = 0; mv = 0; } /* a0 and a1 are read by the ecall chip */
      const int load = cls == CL_LW || cls == CL_LDS, store = cls == CL_SW || cls == CL_STS;
      const uint32_t code = orc_code_of(op);
      T(C_PC) = pc;
      T(SELC(cls)) = 1;
      const int uc = orc_ucmp_of(op);
      T(C_CODE) = code; T(C_UC) = (uint32_t)uc; T(C_WR) = wr; T(C_USE2) = use2; T(C_RD) = rd; T(C_RS1) = rs1; T(C_RS2) = rs2;
      put_limbs(t, h, r, C_IMM_LO, imm); put_limbs(t, h, r, C_TGT_LO, tgt);
      uint32_t x = 0, next = pc + 4, k0 = 0, k1 = 0;
      int off = -1;
      const uint32_t blo = b & 0xffff, bhi = b >> 16, clo = c & 0xffff, chi = c >> 16;
      switch (cls) {
        case CL_ADD: x = a; k0 = (blo + clo) >> 16; k1 = (bhi + chi + k0) >> 16; break;
        case CL_SUB: x = a; k0 = ((a & 0xffff) + clo) >> 16; k1 = ((a >> 16) + chi + k0) >> 16; break;
        case CL_JAL: next = tgt; break;
        case CL_JALR:
          x = b + c; k0 = (blo + clo) >> 16; k1 = (bhi + chi + k0) >> 16;
          off = (int)(x & 1); next = x & ~1u;
          break;
        case CL_LW: case CL_LDS: case CL_SW: case CL_STS: /* the address is rs1 + immediate */
          x = b + imm; k0 = (blo + (imm & 0xffff)) >> 16; k1 = (bhi + (imm >> 16) + k0) >> 16;
          off = (int)(x & 3);
          break;
        case CL_BEQ: case CL_BNE: {
          k0 = blo == clo; k1 = bhi == chi;
          T(C_X) = k0 ? 0 : f_inv(f_sub(blo, clo));
          T(C_X + 1) = k1 ? 0 : f_inv(f_sub(bhi, chi));
          a = k0 & k1;
          if ((cls == CL_BEQ) == (a != 0)) next = tgt;
          break;
        }
        case CL_BLT: case CL_BGE:
          a = less_than(code, b, c);
          if ((cls == CL_BLT) == (a != 0)) next = tgt;
          break;
        case CL_ALU: break; /* (sltu: below) */
        case CL_ECALL:
          x = a;
          if (b == 0x00) next = pad_pc;
          break;
        case CL_KECCAK: x = b; next = b; break;
        default: break;
      }
      if (uc) { /* the row's own unsigned comparison: X = B - C + 2^32 [B < C], limb by limb with borrows K0, K1 */
        k0 = blo < clo;
        k1 = b < c;
        x = b - c;
      }
      if (cls != CL_BEQ && cls != CL_BNE) put_limbs(t, h, r, C_X, x);
      /* second access: rs2, or the word a load reads (which then sits in C); written location: rd, or the word a store
       * leaves behind (in A, its old value in W_P) */
      put_limbs(t, h, r, C_A, store ? mv : a); put_limbs(t, h, r, C_B, b); put_limbs(t, h, r, C_C, load ? m : c);
      T(C_K0) = k0; T(C_K1) = k1;
      if (off > 0) T(C_O1 + off - 1) = 1;
      T(C_NEXT_PC) = next;
      T(C_ADDR2) = use2 ? rs2 : load ? (x & ~3u) : 0;
      T(C_ADDR3) = wr ? rd : store ? (x & ~3u) : 0;
      gap[0] = ts - cy[7] - 1;
      if (use2) gap[1] = ts - cy[8];
      if (load) gap[1] = ts - cy[9];
      if (wr) { gap[2] = ts + 1 - cy[10]; T(C_W_PLO) = wprev & 0xffff; T(C_W_PHI) = wprev >> 16; }
      if (store) { gap[2] = ts + 1 - cy[9]; T(C_W_PLO) = m & 0xffff; T(C_W_PHI) = m >> 16; }
      /* soundness tests: ZKSP_ORACLE_ADDR=<row> moves that row's memory slot (a load reads the next word, anything
       * else writes its result to another register): the address columns are tied to the instruction's register
       * fields or to the adder output by constraints of their own */
      const char* ah = getenv("ZKSP_ORACLE_ADDR");
      if (ah && r == (size_t)strtoull(ah, NULL, 10)) {
        if (load) T(C_ADDR2) += 4;
        else if (wr) T(C_ADDR3) ^= 1;
      }
      /* soundness tests: ZKSP_ORACLE_NONCANON=<row> makes that addition claim the other carry, i.e. write the same
       * sum with limbs out of range.  The adder constraints still hold; the range lookup cannot. */
      const char* nc = getenv("ZKSP_ORACLE_NONCANON");
      if (nc && cls == CL_ADD && r == (size_t)strtoull(nc, NULL, 10)) {
        k0 ^= 1u;
        T(C_K0) = k0;
        T(C_X) = f_sub(f_add(blo, clo), k0 ? 65536u : 0u);
        T(C_X + 1) = f_sub(f_add(f_add(bhi, chi), k0), k1 ? 65536u : 0u);
        T(C_A) = T(C_X); T(C_A + 1) = T(C_X + 1);
      }
    }
    for (int q = 0; q < 3; ++q) { T(C_GAP + 2 * q) = gap[q] & 0xffff; T(C_GAP + 2 * q + 1) = gap[q] >> 16; }
#undef T
  }
}

/* rows [0, h) of an ALU / sub-word instance whose first row is event `row0` */
static void fill_alu(const orc_machine_input* in, size_t h, uint32_t* t, size_t row0) {
  const size_t n = orc_machine_events(in, 0, NULL);
  uint32_t* ev = (uint32_t*)malloc((n ? n : 1) * 4);
  orc_machine_events(in, 0, ev);
  for (size_t r = 0; r < h && row0 + r < n; ++r) {
#define T(col) t[(size_t)(col) * h + r]
    const uint32_t* cy = in->cycles + 12 * (size_t)ev[row0 + r];
    const uint32_t code = orc_code_of(prog_row(in, cy[0])[1]), b = cy[2], c = cy[3];
    uint32_t a = cy[1], x = 0, k0 = 0, k1 = 0;
    if (code == OP_SLL || code == OP_SRL || code == OP_SRA) x = 1u << (c & 31);
    if (code == OP_SLT) {
      /* X = B - C + 2^32 * [less than], the sign bits swapped for the signed order */
      const uint32_t blo = b & 0xffff, bhi = b >> 16, clo = c & 0xffff, chi = c >> 16;
      k0 = blo < clo;
      k1 = less_than(code, b, c);
      const uint32_t dlo = blo - clo + 65536 * k0;
      const int64_t dhi = (int64_t)bhi - chi - k0 + 65536 * (int64_t)k1 + 65536 * ((int64_t)(c >> 31) - (int64_t)(b >> 31));
      x = dlo | ((uint32_t)dhi << 16);
      a = k1; /* branches: the record's a is 0, the flag is what the CPU row sends */
    }
    T(AL_IS_REAL) = 1;
    T(AL_SEL + (code - OP_SLL)) = 1;
    put_limbs(t, h, r, AL_A, a);
    put_bits(t, h, r, AL_B, b, 32); put_bits(t, h, r, AL_C, c, 32); put_bits(t, h, r, AL_X, x, 32);
    T(AL_K0) = k0; T(AL_K1) = k1;
    const char* wa = getenv("ZKSP_ORACLE_WRONG_ALU");
    if (wa && row0 + r == (size_t)strtoull(wa, NULL, 10)) T(AL_A) = (a & 0xffff) ^ 1u; /* soundness tests: a wrong result */
#undef T
  }
  free(ev);
}
/* the ecall chip: row r is the r-th ecall of the run */
static void fill_ecall(const orc_machine_input* in, size_t h, uint32_t* t) {
  const size_t n = orc_machine_events(in, 3, NULL);
  uint32_t* ev = (uint32_t*)malloc((n ? n : 1) * 4);
  orc_machine_events(in, 3, ev);
  const uint32_t pad_pc = pad_pc_of(in);
  for (size_t r = 0; r < h && r < n; ++r) {
#define T(col) t[(size_t)(col) * h + r]
    static const uint32_t codes[6] = {0x00, 0x02, 0x10, 0x1a, 0xf0, 0xf1};
    const uint32_t g = ev[r], ts = 4 * (g + 1);
    const uint32_t* cy = in->cycles + 12 * (size_t)g;
    const uint32_t pc = cy[0], a = cy[1], b = cy[2], c = cy[3], m = cy[4];
    T(EC_IS_REAL) = 1;
    for (int k = 0; k < 6; ++k) if (b == codes[k]) T(EC_SC + k) = 1;
    T(EC_TS) = ts; T(EC_PC) = pc; T(EC_NP) = b == 0x00 ? pad_pc : pc + 4;
    T(EC_B_LO) = b & 0xffff; T(EC_A_LO) = a & 0xffff; T(EC_A_HI) = a >> 16;
    T(EC_C_LO) = c & 0xffff; T(EC_C_HI) = c >> 16; T(EC_M_LO) = m & 0xffff; T(EC_M_HI) = m >> 16;
    const uint32_t g0 = ts - cy[8], g1 = ts + 1 - cy[9];
    T(EC_GAP) = g0 & 0xffff; T(EC_GAP + 1) = g0 >> 16; T(EC_GAP + 2) = g1 & 0xffff; T(EC_GAP + 3) = g1 >> 16;
    if (b == 0xf1) { /* a HINT_READ of m bytes covers NW = ceil(m / 4) words */
      const uint32_t nw = (m + 3) / 4, padb = 4 * nw - m;
      T(EC_NW) = nw; T(EC_P1) = padb & 1u; T(EC_P2) = (padb >> 1) & 1u;
    }
    /* soundness tests: ZKSP_ORACLE_ECALL_NP=<row> sends that (non-HALT) ecall to the padding instruction - the CPU row
     * goes on at pc + 4, so the ECALL bus cannot balance; ZKSP_ORACLE_ECALL_FLAG=<row> decodes that ecall as the next
     * syscall in the list - the code in t0 no longer matches the flags */
    const char* hk = getenv("ZKSP_ORACLE_ECALL_NP");
    if (hk && r == (size_t)strtoull(hk, NULL, 10) && b != 0x00) T(EC_NP) = pad_pc;
    hk = getenv("ZKSP_ORACLE_ECALL_FLAG");
    if (hk && r == (size_t)strtoull(hk, NULL, 10))
      for (int k = 0; k < 6; ++k) if (b == codes[k]) { T(EC_SC + k) = 0; T(EC_SC + (k + 1) % 6) = 1; }
#undef T
  }
  free(ev);
}
/* rows [0, h) of a bitwise instance whose first row is event `row0` */
static void fill_bw(const orc_machine_input* in, size_t h, uint32_t* t, size_t row0) {
  const size_t n = orc_machine_events(in, 2, NULL);
  uint32_t* ev = (uint32_t*)malloc((n ? n : 1) * 4);
  orc_machine_events(in, 2, ev);
  for (size_t r = 0; r < h && row0 + r < n; ++r) {
#define T(col) t[(size_t)(col) * h + r]
    const uint32_t* cy = in->cycles + 12 * (size_t)ev[row0 + r];
    const uint32_t code = orc_code_of(prog_row(in, cy[0])[1]), a = cy[1], b = cy[2], c = cy[3];
    T(BW_IS_REAL) = 1;
    T(BW_SEL + (code - OP_XOR)) = 1;
    for (int i = 0; i < 4; ++i) { T(BW_A + i) = (a >> (8 * i)) & 0xff; T(BW_B + i) = (b >> (8 * i)) & 0xff; T(BW_C + i) = (c >> (8 * i)) & 0xff; }
    const char* wa = getenv("ZKSP_ORACLE_WRONG_BW");
    if (wa && row0 + r == (size_t)strtoull(wa, NULL, 10)) T(BW_A) = ((a & 0xff) ^ 1u); /* soundness tests: a wrong result byte */
#undef T
  }
  free(ev);
}
static int sub_sel(uint32_t code) {
  return code == OP_LB ? 0 : code == OP_LH ? 1 : code == OP_LBU ? 2 : code == OP_LHU ? 3 : code == OP_SB ? 4 : 5;
}
static void fill_sub(const orc_machine_input* in, size_t h, uint32_t* t, size_t row0) {
  const size_t n = orc_machine_events(in, 1, NULL);
  uint32_t* ev = (uint32_t*)malloc((n ? n : 1) * 4);
  orc_machine_events(in, 1, ev);
  for (size_t r = 0; r < h && row0 + r < n; ++r) {
#define T(col) t[(size_t)(col) * h + r]
    const uint32_t* cy = in->cycles + 12 * (size_t)ev[row0 + r];
    const uint32_t* p = prog_row(in, cy[0]);
    const uint32_t code = p[1], a = cy[1], b = cy[2], c = cy[3], m = cy[4], mv = cy[5];
    const int store = code == OP_SB || code == OP_SH;
    const uint32_t off = (b + p[7]) & 3; /* loads: c is the immediate as well */
    T(SW_IS_REAL) = 1;
    T(SW_SEL + sub_sel(code)) = 1;
    T(SW_O + off) = 1;
    put_limbs(t, h, r, SW_A, store ? mv : a); /* loads: the value loaded; stores: the word left behind */
    for (int k = 0; k < 4; ++k) T(SW_MB + k) = (m >> (8 * k)) & 0xff;
    if (store) { T(SW_CB) = c & 0xff; T(SW_CB + 1) = (c >> 8) & 0xff; }
    if (code == OP_LB || code == OP_LH) {
      const uint32_t sb = code == OP_LB ? (m >> (8 * off)) & 0xff : (m >> (8 * (off | 1))) & 0xff;
      T(SW_SELB) = sb;
      T(SW_S) = sb >> 7;
      /* soundness tests: ZKSP_ORACLE_SUB_SIGN=<row> claims the other sign for that signed load (and extends it): the
       * table chip has no byte-operation row (and, byte, 0x80, 128 * sign) for it */
      const char* hk = getenv("ZKSP_ORACLE_SUB_SIGN");
      if (hk && row0 + r == (size_t)strtoull(hk, NULL, 10)) {
        const uint32_t s2 = (sb >> 7) ^ 1u, lo = code == OP_LB ? (sb | (s2 ? 0xff00u : 0u)) : (a & 0xffff);
        T(SW_S) = s2;
        T(SW_A) = lo; T(SW_A + 1) = s2 ? 0xffffu : 0u;
      }
    }
#undef T
  }
  free(ev);
}

/* multiplicities of the table chip: whatever the other chips' rows look up (their receives on the RANGE and BYTES
 * buses, evaluated on their own traces, so that by construction the buses balance when every value is in range) */
static void fill_table_mults(const orc_machine_input* in, uint32_t* t) {
  const size_t ht = (size_t)1 << TABLE_LOG_H;
  int logh[N_CHIPS];
  orc_machine_heights(in, logh);
  static const int users[10 + CPU_INST] = {CH_CPU, CH_CPU2, CH_KMEM, CH_MEMFINAL, CH_BW, CH_BW2, CH_SUB, CH_SUB2, CH_ECALL, CH_P2,
                                           CH_CPU3, CH_CPU4, CH_CPU5, CH_CPU6, CH_CPU7, CH_CPU8, CH_MUL, CH_DIV};
  for (int u = 0; u < 10 + CPU_INST; ++u) {
    const int chip = users[u];
    const orc_chip* ch = &g_chips[chip];
    const size_t h = (size_t)1 << logh[chip];
    uint32_t* tr = (uint32_t*)malloc((size_t)ch->main_width * h * 4);
    orc_machine_fill(in, chip, logh[chip], NULL, tr);
    uint32_t* row = (uint32_t*)malloc((size_t)ch->main_width * 4);
    for (size_t r = 0; r < h; ++r) {
      for (int c = 0; c < ch->main_width; ++c) row[c] = tr[(size_t)c * h + r];
      for (int k = 0; k < ch->n_inter; ++k) {
        const orc_inter* it = &ch->inter[k];
        if (it->sign > 0 || (it->bus != BUS_RANGE && it->bus != BUS_BYTES && it->bus != BUS_BYTEOP)) continue;
        uint32_t v[5] = {0, 0, 0, 0, 0};
        const orc_lf* f[5] = {&it->mult, &it->el[0], &it->el[1], &it->el[2], &it->el[3]};
        for (int j = 0; j < 1 + it->n_el; ++j) {
          v[j] = f[j]->c0;
          for (int i = 0; i < f[j]->n; ++i) v[j] = f_add(v[j], f_mul(f[j]->coef[i], row[f[j]->col[i]]));
        }
        if (v[0] == 0) continue;
        if (it->bus == BUS_RANGE) {
          if (v[2] >= ht || v[1] > 2 || (v[1] == 1 && (v[2] & 3)) || (v[1] == 2 && (v[2] == 0 || v[2] > ADDR_HI_MAX))) continue; /* no table row: the buses will not balance */
          uint32_t* dst = &t[(size_t)(v[1] == 0 ? TB_M_R16 : v[1] == 1 ? TB_M_AL : TB_M_TOP) * ht + v[2]];
          *dst = f_add(*dst, v[0]);
        } else if (it->bus == BUS_BYTES) {
          if (v[1] > 255 || v[2] > 255) continue;
          uint32_t* dst = &t[(size_t)TB_M_BY * ht + v[1] + 256 * v[2]];
          *dst = f_add(*dst, v[0]);
        } else { /* (kind, x, y, z): counted only if z is the table's answer */
          if (v[1] < 1 || v[1] > 3 || v[2] > 255 || v[3] > 255) continue;
          const uint32_t z = v[1] == 1 ? (v[2] ^ v[3]) : v[1] == 2 ? (v[2] | v[3]) : (v[2] & v[3]);
          if (v[4] != z) continue;
          uint32_t* dst = &t[(size_t)(TB_M_XOR + v[1] - 1) * ht + v[2] + 256 * v[3]];
          *dst = f_add(*dst, v[0]);
        }
      }
    }
    free(row);
    free(tr);
  }
}

void orc_machine_fill(const orc_machine_input* in, int chip, int logh, uint32_t* prep, uint32_t* t) {
  build();
  const size_t h = (size_t)1 << logh;
  const orc_chip* ch = &g_chips[chip];
  memset(t, 0, (size_t)ch->main_width * h * 4);
  if (prep) memset(prep, 0, (size_t)ch->prep_width * h * 4);
#define T(col) t[(size_t)(col) * h + r]
  switch (chip) {
    case CH_CPU: case CH_CPU2: case CH_CPU3: case CH_CPU4: case CH_CPU5: case CH_CPU6: case CH_CPU7: case CH_CPU8: {
      int lh[N_CHIPS];
      orc_machine_heights(in, lh);
      fill_cpu(in, h, t, cpu_row0(lh, orc_cpu_instance(chip)));
      break;
    }
    case CH_ALU: fill_alu(in, h, t, 0); break;
    case CH_ALU2: fill_alu(in, h, t, first_rows(in, CH_ALU, orc_machine_events(in, 0, NULL))); break;
    case CH_SUB: fill_sub(in, h, t, 0); break;
    case CH_SUB2: fill_sub(in, h, t, first_rows(in, CH_SUB, orc_machine_events(in, 1, NULL))); break;
    case CH_ECALL: fill_ecall(in, h, t); break;
    case CH_BW: fill_bw(in, h, t, 0); break;
    case CH_BW2: fill_bw(in, h, t, first_rows(in, CH_BW, orc_machine_events(in, 2, NULL))); break;
    case CH_KECCAK: {
      uint64_t* st = (uint64_t*)calloc(25 * (in->n_keccak ? in->n_keccak : 1), 8);
      for (size_t p = 0; p < in->n_keccak; ++p) memcpy(st + 25 * p, ((const kcall_t*)(in->keccak + 408 * p))->in, 200);
      orc_keccak_trace(st, (int)in->n_keccak, logh, t); /* fills columns 0..KA_WIDTH-1 of a [.][h] matrix */
      free(st);
      for (size_t p = 0; p < in->n_keccak; ++p)
        for (int rr = 0; rr < 24; ++rr) { size_t r = 24 * p + rr; T(KC_TS) = ((const kcall_t*)(in->keccak + 408 * p))->ts; }
      break;
    }
    case CH_KMEM:
      for (size_t r = 0; r < h; ++r) {
        const size_t p = r / 50, i = r % 50;
        T(KM_IDX) = (uint32_t)i; T(KM_ISF) = i == 0; T(KM_ISL) = i == 49;
        if (p >= in->n_keccak) continue;
        const kcall_t* k = (const kcall_t*)(in->keccak + 408 * p);
        uint64_t o[25];
        memcpy(o, k->in, 200);
        orc_keccak_f(o);
        const uint32_t wi = (uint32_t)(k->in[i >> 1] >> (32 * (i & 1))), wo = (uint32_t)(o[i >> 1] >> (32 * (i & 1)));
        T(KM_IS_REAL) = 1; T(KM_TS) = k->ts; T(KM_PTR_LO) = k->ptr & 0xffff; T(KM_PTR_HI) = k->ptr >> 16;
        T(KM_CALL) = i == 0; T(KM_ADDR) = k->ptr + 4 * (uint32_t)i;
        T(KM_OLD_LO) = wi & 0xffff; T(KM_OLD_HI) = wi >> 16; T(KM_NEW_LO) = wo & 0xffff; T(KM_NEW_HI) = wo >> 16;
        T(KM_PTS) = k->pts[i];
        const uint32_t gap = k->ts + 1 - k->pts[i];
        T(KM_GL) = gap & 0xffff; T(KM_GH) = gap >> 16;
      }
      break;
    case CH_MEMFINAL: {
      /* soundness tests: ZKSP_ORACLE_MF_WRAP=<row> claims the difference to the next row's address that only holds
       * mod p when that address does not increase (DIFF = p - 1 for a repeated address, as limbs) */
      const char* wrap = getenv("ZKSP_ORACLE_MF_WRAP");
      const size_t crows = in->n_cycles ? cpu_rows(in) : 0;
      for (size_t r = 0; r < in->n_memfinal; ++r) {
        const uint32_t* f = in->memfinal + 5 * r;
        T(MF_IS_REAL) = 1; T(MF_LO) = f[0] & 0xffff; T(MF_HI) = f[0] >> 16; T(MF_IS_INIT) = f[4] == 1; T(MF_IS_ZERO) = f[4] == 2;
        T(MF_INIT_LO) = f[1] & 0xffff; T(MF_INIT_HI) = f[1] >> 16;
        T(MF_FIN_LO) = f[2] & 0xffff; T(MF_FIN_HI) = f[2] >> 16; T(MF_FIN_TS) = f[3];
        /* x0 (row 0) is read once more by every CPU row after the last cycle: its last access is the last row's */
        if (r == 0 && crows > in->n_cycles) T(MF_FIN_TS) = 4 * (uint32_t)crows;
        if (r + 1 < in->n_memfinal) {
          const uint32_t nx = in->memfinal[5 * (r + 1)];
          const uint32_t bw = (nx & 0xffff) < (f[0] & 0xffff) + 1;
          T(MF_BW) = bw;
          T(MF_D_LO) = (nx & 0xffff) + 65536 * bw - (f[0] & 0xffff) - 1;
          T(MF_D_HI) = f_sub(f_sub(nx >> 16, f[0] >> 16), bw);
          if (wrap && r == (size_t)strtoull(wrap, NULL, 10)) {
            const uint32_t d = f_sub(f_sub(nx % FP, f[0] % FP), 1); /* the difference mod p */
            T(MF_BW) = 0; T(MF_D_LO) = d & 0xffff; T(MF_D_HI) = d >> 16;
          }
        }
      }
      break;
    }
    case CH_IMAGE: {
      /* soundness tests: ZKSP_ORACLE_IMG_UNUSED=<row> withholds that image word from the IMG bus (the round-2 attack: the
       * word is then free-initialised by the memory-boundary chip); the chip's constraint USED = is_real forbids it */
      const char* unused = getenv("ZKSP_ORACLE_IMG_UNUSED");
      for (size_t r = 0; r < in->n_image; ++r) {
        prep[(size_t)IMG_P_ADDR * h + r] = in->image[2 * r];
        prep[(size_t)IMG_P_LO * h + r] = in->image[2 * r + 1] & 0xffff;
        prep[(size_t)IMG_P_HI * h + r] = in->image[2 * r + 1] >> 16;
        prep[(size_t)IMG_P_REAL * h + r] = 1;
        T(0) = !(unused && r == (size_t)strtoull(unused, NULL, 10));
      }
      break;
    }
    case CH_PROGRAM:
      for (size_t r = 0; r < in->n_program; ++r) {
        const uint32_t* p = in->program + 9 * r;
        uint32_t* q = prep + r;
        q[(size_t)PR_PC * h] = p[0]; q[(size_t)PR_CLS * h] = (uint32_t)orc_class_of(p[1]); q[(size_t)PR_CODE * h] = orc_code_of(p[1]);
        q[(size_t)PR_UC * h] = (uint32_t)orc_ucmp_of(p[1]);
        const int ecall = p[1] == OP_ECALL; /* its CPU row moves t0 only: a0 and a1 are the ecall chip's reads */
        q[(size_t)PR_WR * h] = p[2]; q[(size_t)PR_USE2 * h] = ecall ? 0 : p[3];
        q[(size_t)PR_RD * h] = p[4]; q[(size_t)PR_RS1 * h] = p[5]; q[(size_t)PR_RS2 * h] = ecall ? 0 : p[6];
        q[(size_t)PR_IMM_LO * h] = p[7] & 0xffff; q[(size_t)PR_IMM_HI * h] = p[7] >> 16;
        q[(size_t)PR_TGT_LO * h] = p[8] & 0xffff; q[(size_t)PR_TGT_HI * h] = p[8] >> 16;
        if (in->prog_mult) T(0) = in->prog_mult[r] % FP;
      }
      /* the padding instruction (last row) is fetched by every CPU row after the last cycle */
      if (in->prog_mult && in->n_cycles) { const size_t r = in->n_program - 1; T(0) = (uint32_t)(cpu_rows(in) - in->n_cycles); }
      break;
    case CH_MUL:
      for (size_t r = 0; r < in->n_muls; ++r) {
        const uint32_t* mu = in->muls + 3 * r;
        const uint32_t b = mu[1], c = mu[2];
        const uint64_t prod = (uint64_t)b * c;
        T(MU_IS_REAL) = 1; T(MU_HI) = mu[0] == 1; T(MU_SH) = mu[0] == 2; T(MU_SHU) = mu[0] == 3;
        if (mu[0] >= 2) { /* mulh / mulhsu: R + b31 * C + [mulh] c31 * B = P_hi + 2^32 k, limb by limb */
          const uint32_t b31 = b >> 31, c31 = mu[0] == 2 ? c >> 31 : 0, phi = (uint32_t)(((uint64_t)b * c) >> 32);
          const uint32_t rr = phi - (b31 ? c : 0) - (c31 ? b : 0);
          const uint32_t lo = (rr & 0xffff) + (b31 ? (c & 0xffff) : 0) + (c31 ? (b & 0xffff) : 0);
          const uint32_t k0 = (lo - (phi & 0xffff)) >> 16;
          const uint32_t hi = (rr >> 16) + (b31 ? (c >> 16) : 0) + (c31 ? (b >> 16) : 0) + k0;
          const uint32_t k1 = (hi - (phi >> 16)) >> 16;
          put_limbs(t, h, r, MU_R, rr);
          T(MU_K0) = k0 >= 1; T(MU_K0 + 1) = k0 >= 2; T(MU_K1) = k1 >= 1; T(MU_K1 + 1) = k1 >= 2;
        }
        put_bits(t, h, r, MU_B, b, 32); put_bits(t, h, r, MU_C, c, 32);
        put_bits(t, h, r, MU_P, (uint32_t)prod, 32); put_bits(t, h, r, MU_P + 32, (uint32_t)(prod >> 32), 32);
        uint64_t s[7] = {0};
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j) s[i + j] += (uint64_t)((b >> (8 * i)) & 0xff) * ((c >> (8 * j)) & 0xff);
        const uint64_t q0 = (s[0] + 256 * s[1]) >> 16, q1 = (s[2] + 256 * s[3] + q0) >> 16, q2 = (s[4] + 256 * s[5] + q1) >> 16;
        put_bits(t, h, r, MU_Q0, (uint32_t)q0, 10); put_bits(t, h, r, MU_Q1, (uint32_t)q1, 11); put_bits(t, h, r, MU_Q2, (uint32_t)q2, 10);
      }
      break;
    case CH_P2: {
      /* the aggregation payload's node rows (one per ancestor of a supplied key, ascending: the node's key, its children's
       * digests in, its own out), then the rows of a leaf-proof check as recorded */
      size_t nr = orc_machine_agg_rows(in->agg_keys, in->agg_leaves, in->n_agg, NULL);
      if (nr == (size_t)-1) nr = 0;
      uint32_t* rows = (uint32_t*)calloc(25 * (nr ? nr : 1), 4);
      if (nr) orc_machine_agg_rows(in->agg_keys, in->agg_leaves, in->n_agg, rows);
      uint32_t ext_rc[8][16], int_rc[13];
      orc_poseidon2_constants(&ext_rc[0][0], int_rc);
      for (size_t r = 0; r < h; ++r) {
        uint32_t st[16] = {0};
        if (r < nr) {
          T(P2_IS_REAL) = 1; T(P2_KL) = rows[25 * r] & 0xffff; T(P2_KH) = rows[25 * r] >> 16; T(P2_FN) = 1;
          memcpy(st, rows + 25 * r + 1, 64);
        } else if (r < nr + in->n_leaf_p2) {
          const uint32_t* rc = in->leaf_p2_rows + P2_REC_WORDS * (r - nr);
          const uint32_t kind = rc[0] & 15u;
          T(P2_IS_REAL) = 1; T(P2_T) = rc[1] % FP; T(P2_KL) = rc[2] & 0xffff; T(P2_KH) = rc[2] >> 16; T(P2_M) = rc[3] % FP;
          T(P2_FN) = kind == P2K_NODE; T(P2_SZ) = kind == P2K_SZ; T(P2_SC) = kind == P2K_SC; T(P2_PL) = kind == P2K_PL;
          T(P2_PR) = kind == P2K_PR; T(P2_FJ) = kind == P2K_J;
          T(P2_NEW) = (rc[0] & P2F_NEW) != 0; T(P2_SND) = (rc[0] & P2F_SND) != 0; T(P2_FR) = (rc[0] & P2F_FRI) != 0;
          T(P2_RE) = (rc[0] & P2F_RE) != 0; T(P2_SE) = (rc[0] & P2F_SE) != 0; T(P2_RID) = rc[P2_REC_RID] % FP;
          fe4 ap;
          for (int i = 0; i < 4; ++i) { T(P2_SO + i) = rc[P2_REC_SO + i] % FP; ap.c[i] = rc[P2_REC_ALPHA + i] % FP; }
          fe4 pw = ap; /* alpha^1 .. alpha^8 */
          for (int j = 0; j < 8; ++j) {
            for (int i = 0; i < 4; ++i) T(P2_AP + 4 * j + i) = pw.c[i];
            pw = e_mul(pw, ap);
          }
          for (int i = 0; i < 16; ++i) st[i] = rc[4 + i] % FP;
        }
        for (int i = 0; i < 16; ++i) T(P2_IN + i) = st[i];
        orc_p2_external_linear(st);
        for (int rd = 0; rd < 8; ++rd) {
          if (rd == 4) /* the 13 internal rounds sit between the two halves of the external ones */
            for (int ir = 0; ir < 13; ++ir) {
              const fe x = f_add(st[0], int_rc[ir]), x3 = f_mul(f_mul(x, x), x), y = f_mul(f_mul(x3, x3), x);
              T(P2_INT + 2 * ir) = x3; T(P2_INT + 2 * ir + 1) = y;
              st[0] = y;
              orc_p2_internal_linear(st);
            }
          for (int i = 0; i < 16; ++i) {
            const fe x = f_add(st[i], ext_rc[rd][i]), x3 = f_mul(f_mul(x, x), x), y = f_mul(f_mul(x3, x3), x);
            T(P2_EXT + 32 * rd + i) = x3; T(P2_EXT + 32 * rd + 16 + i) = y;
            st[i] = y;
          }
          orc_p2_external_linear(st);
        }
        if (r < nr && memcmp(st, rows + 25 * r + 17, 32) != 0) abort(); /* the row's permutation is the node's compression */
      }
      free(rows);
      break;
    }
    case CH_DIV: {
      /* one row per div / divu / rem / remu cycle, in execution order (the oracle's own event list); the row holds the
       * instruction's result computed from the operands - a cycle record claiming another one leaves the ALU bus open */
      const size_t nd = orc_machine_events(in, 4, NULL);
      uint32_t* ev = (uint32_t*)malloc((nd ? nd : 1) * 4);
      orc_machine_events(in, 4, ev);
      for (size_t r = 0; r < nd && r < h; ++r) {
        const uint32_t* cy = in->cycles + 12 * (size_t)ev[r];
        const uint32_t code = prog_row(in, cy[0])[1], n = cy[2], d = cy[3];
        const int sg = code == OP_DIV || code == OP_REM, ovf = sg && n == 0x80000000u && d == 0xffffffffu;
        uint32_t q, rm;
        if (d == 0) { q = 0xffffffffu; rm = n; }
        else if (ovf) { q = n; rm = 0; }
        else if (sg) { q = (uint32_t)((int32_t)n / (int32_t)d); rm = (uint32_t)((int32_t)n % (int32_t)d); }
        else { q = n / d; rm = n % d; }
        const uint32_t sn = sg ? n >> 31 : 0, sd = sg ? d >> 31 : 0;
        const uint32_t an = sn ? 0u - n : n, ad = sd ? 0u - d : d;
        uint32_t sq, sr, aq, ar;
        if (d == 0) { sq = 1; aq = 1; sr = sn; ar = an; }
        else { aq = an / ad; ar = an % ad; sq = (sn ^ sd) & (aq != 0); sr = sn & (ar != 0); }
        T(DV_IS_REAL) = 1;
        T(DV_F + 0) = code == OP_DIV; T(DV_F + 1) = code == OP_DIVU; T(DV_F + 2) = code == OP_REM; T(DV_F + 3) = code == OP_REMU;
        put_limbs(t, h, r, DV_N, n); put_limbs(t, h, r, DV_D, d);
        put_limbs(t, h, r, DV_A, (code == OP_DIV || code == OP_DIVU) ? q : rm);
        T(DV_SN) = sn; T(DV_SD) = sd; T(DV_NH) = (n >> 16) - 32768u * sn; T(DV_DH) = (d >> 16) - 32768u * sd;
        put_limbs(t, h, r, DV_AN, an); put_limbs(t, h, r, DV_AD, ad); put_limbs(t, h, r, DV_AQ, aq); put_limbs(t, h, r, DV_AR, ar);
        T(DV_CN) = sn && (n & 0xffff) != 0; T(DV_CD) = sd && (d & 0xffff) != 0;
        T(DV_CQ) = sq && (q & 0xffff) != 0; T(DV_CR) = sr && (rm & 0xffff) != 0;
        put_limbs(t, h, r, DV_Q, q); put_limbs(t, h, r, DV_R, rm);
        T(DV_SQ) = sq; T(DV_SR) = sr; T(DV_XS) = sn ^ sd;
        const uint32_t pl = d == 0 ? 0 : aq * ad, e = d == 0 ? 0 : ad - ar - 1;
        put_limbs(t, h, r, DV_PL, pl);
        T(DV_K) = d != 0 && (pl & 0xffff) + (ar & 0xffff) > 0xffff;
        put_limbs(t, h, r, DV_E, e);
        T(DV_BE) = d != 0 && (ad & 0xffff) < (ar & 0xffff) + 1;
        T(DV_NZD) = d != 0; T(DV_INVD) = d ? f_inv(((d & 0xffff) + (d >> 16)) % FP) : 0;
        T(DV_NZQ) = aq != 0; T(DV_INVQ) = aq ? f_inv(((aq & 0xffff) + (aq >> 16)) % FP) : 0;
        T(DV_NZR) = ar != 0; T(DV_INVR) = ar ? f_inv(((ar & 0xffff) + (ar >> 16)) % FP) : 0;
      }
      free(ev);
      break;
    }
    case CH_HINT: {
      /* one row per word of every HINT_READ, in execution order of the reads; USED and the value from the memory boundary list */
      const size_t ne = orc_machine_events(in, 3, NULL);
      uint32_t* ev = (uint32_t*)malloc((ne ? ne : 1) * 4);
      orc_machine_events(in, 3, ev);
      size_t r = 0;
      for (size_t e = 0; e < ne; ++e) {
        const uint32_t* cy = in->cycles + 12 * (size_t)ev[e];
        if (cy[2] != 0xf1) continue;
        const uint32_t ptr = cy[3], nw = (cy[4] + 3) / 4;
        for (uint32_t j = 0; j < nw && r < h; ++j, ++r) {
          const uint32_t addr = ptr + 4 * j;
          size_t lo = 0, hi = in->n_memfinal;
          while (lo < hi) {
            const size_t mid = (lo + hi) >> 1;
            if (in->memfinal[5 * mid] < addr) lo = mid + 1; else hi = mid;
          }
          const int used = lo < in->n_memfinal && in->memfinal[5 * lo] == addr;
          const uint32_t v = used ? in->memfinal[5 * lo + 1] : 0;
          T(HN_IS_REAL) = 1; T(HN_FIRST) = j == 0; T(HN_LAST) = j + 1 == nw; T(HN_ADDR) = addr % FP; T(HN_CNT) = nw - j;
          T(HN_LO) = v & 0xffff; T(HN_HI) = v >> 16; T(HN_USED) = used;
        }
      }
      free(ev);
      break;
    }
    case CH_QR:
      /* 31 rows per query of a leaf-proof check: the record IS the row (the product's verifier computed it while it checked
       * the query; check_constraints and the buses hold it to the AIR) */
      for (size_t r = 0; r < in->n_leaf_qr && r < h; ++r) {
        const uint32_t* rc = in->leaf_qr_rows + QR_REC_WORDS * r;
        for (int c = 0; c < QR_WIDTH; ++c) T(c) = rc[c] % FP;
      }
      break;
    case CH_TR: {
      /* one duplex of a checked leaf's transcript per row, as recorded; the permutation's columns are filled in here */
      uint32_t ext_rc[8][16], int_rc[13];
      orc_poseidon2_constants(&ext_rc[0][0], int_rc);
      for (size_t r = 0; r < h; ++r) {
        uint32_t st[16] = {0};
        if (r < in->n_leaf_tr) {
          const uint32_t* rc = in->leaf_tr_rows + TR_REC_WORDS * r;
          T(TR_IS_REAL) = 1; T(TR_LEAF) = rc[1] % FP; T(TR_STEP) = rc[2] % FP;
          T(TR_FIRST) = (rc[0] >> 16) & 1u; T(TR_ABS) = (rc[0] >> 17) & 1u;
          for (int k = 0; k < 7; ++k) T(TR_UROOT + k) = (rc[0] >> k) & 1u;
          for (int k = 0; k < 8; ++k) T(TR_QM + k) = (rc[0] >> (7 + k)) & 1u;
          T(TR_RIDK) = rc[3] % FP; T(TR_QBASE) = rc[4] % FP;
          for (int k = 0; k < 5; ++k) T(TR_MROOT + k) = rc[5 + k] % FP;
          for (int i = 0; i < 16; ++i) st[i] = rc[10 + i] % FP;
        }
        for (int i = 0; i < 16; ++i) T(TR_IN + i) = st[i];
        orc_p2_external_linear(st);
        for (int rd = 0; rd < 8; ++rd) {
          if (rd == 4)
            for (int ir = 0; ir < 13; ++ir) {
              const fe x = f_add(st[0], int_rc[ir]), x3 = f_mul(f_mul(x, x), x), y = f_mul(f_mul(x3, x3), x);
              T(TR_INT + 2 * ir) = x3; T(TR_INT + 2 * ir + 1) = y;
              st[0] = y;
              orc_p2_internal_linear(st);
            }
          for (int i = 0; i < 16; ++i) {
            const fe x = f_add(st[i], ext_rc[rd][i]), x3 = f_mul(f_mul(x, x), x), y = f_mul(f_mul(x3, x3), x);
            T(TR_EXT + 32 * rd + i) = x3; T(TR_EXT + 32 * rd + 16 + i) = y;
            st[i] = y;
          }
          orc_p2_external_linear(st);
        }
      }
      break;
    }
    case CH_TABLE:
      for (size_t r = 0; r < h; ++r) {
        prep[(size_t)TB_P_X * h + r] = (uint32_t)(r & 255);
        prep[(size_t)TB_P_Y * h + r] = (uint32_t)(r >> 8);
        prep[(size_t)TB_P_NA * h + r] = (r & 3) != 0;
        prep[(size_t)TB_P_NT * h + r] = r == 0 || r > ADDR_HI_MAX;
        prep[(size_t)TB_P_XOR * h + r] = (uint32_t)((r & 255) ^ (r >> 8));
        prep[(size_t)TB_P_AND * h + r] = (uint32_t)((r & 255) & (r >> 8));
      }
      if (in->n_cycles) fill_table_mults(in, t); /* setup passes no cycles: only the preprocessed columns are wanted */
      break;
  }
#undef T
}

/* ------------------------------------------------------------------------------------------
 * constraints
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t* out; int k; } sink;
static inline void emit(sink* s, fe v) {
  if (s->out) s->out[s->k] = v;
  s->k++;
}
static inline fe bool_c(fe v) { return f_mul(v, f_sub(v, 1)); }
static inline fe limb_of(const uint32_t* row, int bits, int limb) {
  fe s = 0;
  for (int i = 15; i >= 0; --i) s = f_add(f_add(s, s), row[bits + 16 * limb + i]);
  return s;
}
static inline fe byte_of(const uint32_t* row, int bits, int byte) {
  fe s = 0;
  for (int i = 7; i >= 0; --i) s = f_add(f_add(s, s), row[bits + 8 * byte + i]);
  return s;
}
static inline fe bits_val(const uint32_t* row, int bits, int n) {
  fe s = 0;
  for (int i = n - 1; i >= 0; --i) s = f_add(f_add(s, s), row[bits + i]);
  return s;
}
#define F65536 65536u
static inline fe word_of(const uint32_t* row, int col) { return f_add(row[col], f_mul(F65536, row[col + 1])); }

static void cpu_constraints(const uint32_t* l, const uint32_t* n, fe is_first, fe is_last, fe is_trans, const uint32_t* pub,
                            sink* s) {
  const fe one = 1;
#define S(cls) l[SELC(cls)]
  /* ---- booleans: class selectors, carries, byte offset (WR, USE2 are Program-table values) ---- */
  fe selsum = 0;
  for (int k = 0; k < N_CLS; ++k) { emit(s, bool_c(l[C_SEL + k])); selsum = f_add(selsum, l[C_SEL + k]); }
  emit(s, bool_c(l[C_K0])); emit(s, bool_c(l[C_K1]));
  for (int i = 0; i < 3; ++i) emit(s, bool_c(l[C_O1 + i]));
  /* ---- row structure: exactly one class; the clock; the chain of pcs; the instance's first and last rows ---- */
  emit(s, f_sub(selsum, one));
  emit(s, f_mul(is_first, f_sub(l[C_PC], pub[CPUPUB_START_PC] % FP)));
  emit(s, f_mul(is_first, f_sub(l[C_TS], pub[CPUPUB_START_TS] % FP)));
  emit(s, f_mul(is_trans, f_sub(f_sub(n[C_TS], l[C_TS]), 4)));
  emit(s, f_mul(is_trans, f_sub(n[C_PC], l[C_NEXT_PC])));
  {
    const fe succ = pub[CPUPUB_HAS_SUCC] % FP;
    emit(s, f_mul(f_mul(is_last, succ), f_sub(l[C_NEXT_PC], pub[CPUPUB_END_PC] % FP)));
    emit(s, f_mul(f_mul(is_last, f_sub(one, succ)), f_sub(l[C_NEXT_PC], pub[CPUPUB_PAD_PC] % FP)));
  }
  const fe a_lo = l[C_A], a_hi = l[C_A + 1], b_lo = l[C_B], b_hi = l[C_B + 1], c_lo = l[C_C], c_hi = l[C_C + 1];
  const fe x_lo = l[C_X], x_hi = l[C_X + 1];
  const fe k0 = l[C_K0], k1 = l[C_K1], imm_lo = l[C_IMM_LO], imm_hi = l[C_IMM_HI];
  /* ---- operand C is the immediate, unless it is reg[rs2] or the word a load reads ---- */
  const fe loadw = f_add(S(CL_LW), S(CL_LDS)), storew = f_add(S(CL_SW), S(CL_STS));
  {
    const fe immc = f_sub(f_sub(one, l[C_USE2]), loadw);
    emit(s, f_mul(immc, f_sub(c_lo, imm_lo)));
    emit(s, f_mul(immc, f_sub(c_hi, imm_hi)));
  }
  /* ---- the adder: X = B + C (add, jalr), X = B + imm (loads, stores), X + C = B (sub) ---- */
  {
    const fe addc = f_add(S(CL_ADD), S(CL_JALR)), addi = f_add(loadw, storew);
    emit(s, f_mul(addc, f_sub(f_add(b_lo, c_lo), f_add(x_lo, f_mul(F65536, k0)))));
    emit(s, f_mul(addc, f_sub(f_add(f_add(b_hi, c_hi), k0), f_add(x_hi, f_mul(F65536, k1)))));
    emit(s, f_mul(addi, f_sub(f_add(b_lo, imm_lo), f_add(x_lo, f_mul(F65536, k0)))));
    emit(s, f_mul(addi, f_sub(f_add(f_add(b_hi, imm_hi), k0), f_add(x_hi, f_mul(F65536, k1)))));
    emit(s, f_mul(S(CL_SUB), f_sub(f_add(x_lo, c_lo), f_add(b_lo, f_mul(F65536, k0)))));
    emit(s, f_mul(S(CL_SUB), f_sub(f_add(f_add(x_hi, c_hi), k0), f_add(b_hi, f_mul(F65536, k1)))));
    /* what the range lookups check is X: the sum / difference written (add, sub), the value an ecall leaves in t0
     * (HINT_LEN: prover-supplied), the return address of a keccak call */
    const fe cpa = f_add(f_add(S(CL_ADD), S(CL_SUB)), S(CL_ECALL));
    emit(s, f_mul(cpa, f_sub(a_lo, x_lo)));
    emit(s, f_mul(cpa, f_sub(a_hi, x_hi)));
    emit(s, f_mul(S(CL_KECCAK), f_sub(x_lo, b_lo)));
    emit(s, f_mul(S(CL_KECCAK), f_sub(x_hi, b_hi)));
    /* unsigned comparison in the row (sltu, bltu, bgeu): X = B - C + 2^32 K1 limb by limb, so K1 = [B < C] because both
     * limbs of X are looked up in the range table; the flag goes where the ALU chip's answer would */
    const fe uc = l[C_UC];
    emit(s, f_mul(uc, f_sub(f_add(f_sub(b_lo, c_lo), f_mul(F65536, k0)), x_lo)));
    emit(s, f_mul(uc, f_sub(f_add(f_sub(f_sub(b_hi, c_hi), k0), f_mul(F65536, k1)), x_hi)));
    emit(s, f_mul(uc, f_sub(a_lo, k1)));
    emit(s, f_mul(uc, a_hi));
  }
  /* ---- byte offset and the word address ---- */
  const fe o1 = l[C_O1], o2 = l[C_O2], o3 = l[C_O3], osum = f_add(o1, f_add(o2, o3));
  const fe off = f_add(o1, f_add(f_add(o2, o2), f_mul(3, o3)));
  const fe xaddr = f_sub(f_add(x_lo, f_mul(F65536, x_hi)), off);
  {
    const fe noff = f_add(f_add(f_add(f_add(S(CL_ADD), S(CL_SUB)), f_add(S(CL_ECALL), S(CL_KECCAK))), f_add(S(CL_LW), S(CL_SW))), l[C_UC]);
    emit(s, f_mul(noff, osum));
    emit(s, bool_c(osum)); /* at most one of the three offset flags */
    emit(s, f_mul(S(CL_JALR), f_add(o2, o3)));
    /* the second access is rs2 or a load's word, the written location rd or a store's word */
    emit(s, f_mul(l[C_USE2], f_sub(l[C_ADDR2], l[C_RS2])));
    emit(s, f_mul(loadw, f_sub(l[C_ADDR2], xaddr)));
    emit(s, f_mul(l[C_WR], f_sub(l[C_ADDR3], l[C_RD])));
    emit(s, f_mul(storew, f_sub(l[C_ADDR3], xaddr)));
  }
  /* ---- next pc ---- */
  {
    const fe pc4 = f_add(l[C_PC], 4), np = l[C_NEXT_PC], tgt = word_of(l, C_TGT_LO);
    fe def = one;
    const int nd[9] = {CL_JAL, CL_JALR, CL_BEQ, CL_BNE, CL_BLT, CL_BGE, CL_KECCAK, CL_ECALL, 0};
    for (int i = 0; i < 8; ++i) def = f_sub(def, S(nd[i]));
    emit(s, f_mul(def, f_sub(np, pc4)));
    emit(s, f_mul(S(CL_JAL), f_sub(np, tgt)));
    emit(s, f_mul(S(CL_JAL), f_sub(a_lo, c_lo)));
    emit(s, f_mul(S(CL_JAL), f_sub(a_hi, c_hi)));
    emit(s, f_mul(S(CL_JALR), f_sub(a_lo, l[C_TGT_LO])));
    emit(s, f_mul(S(CL_JALR), f_sub(a_hi, l[C_TGT_HI])));
    emit(s, f_mul(S(CL_JALR), f_sub(np, xaddr)));
    /* beq / bne: a limb difference is zero or has the inverse X holds; the flag is "both limbs equal" */
    const fe bq = f_add(S(CL_BEQ), S(CL_BNE)), d_lo = f_sub(b_lo, c_lo), d_hi = f_sub(b_hi, c_hi);
    emit(s, f_mul(bq, f_add(f_sub(f_mul(d_lo, x_lo), one), k0)));
    emit(s, f_mul(bq, f_mul(d_lo, k0)));
    emit(s, f_mul(bq, f_add(f_sub(f_mul(d_hi, x_hi), one), k1)));
    emit(s, f_mul(bq, f_mul(d_hi, k1)));
    emit(s, f_mul(bq, f_sub(a_lo, f_mul(k0, k1))));
    const fe brs = f_add(bq, f_add(S(CL_BLT), S(CL_BGE)));
    emit(s, f_mul(brs, a_hi));
    /* taken -> tgt, else pc + 4 */
    const fe d = f_sub(tgt, pc4), base = f_sub(np, pc4);
    emit(s, f_mul(f_add(S(CL_BEQ), S(CL_BLT)), f_sub(base, f_mul(a_lo, d))));
    emit(s, f_mul(f_add(S(CL_BNE), S(CL_BGE)), f_sub(base, f_mul(f_sub(one, a_lo), d))));
    emit(s, f_mul(S(CL_KECCAK), f_sub(np, f_add(b_lo, f_mul(F65536, b_hi)))));
    /* (ecall: the ecall chip decides the next pc - the next instruction, or the padding instruction after HALT) */
  }
  /* ---- word loads and stores: the register gets the word read (C), the memory the register's value (C) ---- */
  {
    const fe mov = f_add(S(CL_LW), S(CL_SW));
    emit(s, f_mul(mov, f_sub(a_lo, c_lo)));
    emit(s, f_mul(mov, f_sub(a_hi, c_hi)));
  }
  /* ---- ecall: the code in t0 is a 16-bit value (decoded, and the value left behind checked, by the ecall chip) ---- */
  emit(s, f_mul(S(CL_ECALL), b_hi));
  /* previous access times are older by construction: a slot's previous time IS its time - 1 - difference (a linear
   * form in the memory-bus tuples), the difference's low limb and high byte are looked up in the table chip */
#undef S
}

static void kmem_constraints(const uint32_t* l, const uint32_t* n, fe is_first, fe is_trans, sink* s) {
  const fe one = 1;
  emit(s, bool_c(l[KM_IS_REAL])); emit(s, bool_c(l[KM_ISF])); emit(s, bool_c(l[KM_ISL]));
  emit(s, f_sub(l[KM_CALL], f_mul(l[KM_ISF], l[KM_IS_REAL])));
  emit(s, f_mul(is_first, l[KM_IDX]));
  emit(s, f_mul(is_first, f_sub(l[KM_ISF], one)));
  const fe nl = f_sub(one, l[KM_ISL]);
  emit(s, f_mul(is_trans, f_sub(n[KM_IDX], f_mul(f_add(l[KM_IDX], one), nl))));
  emit(s, f_mul(l[KM_ISL], f_sub(l[KM_IDX], 49)));
  emit(s, f_mul(is_trans, f_sub(n[KM_ISF], l[KM_ISL])));
  emit(s, f_mul(f_mul(is_trans, nl), f_sub(n[KM_IS_REAL], l[KM_IS_REAL])));
  emit(s, f_mul(f_mul(is_trans, n[KM_IS_REAL]), f_sub(one, l[KM_IS_REAL])));
  emit(s, f_mul(f_mul(is_trans, nl), f_sub(n[KM_TS], l[KM_TS])));
  emit(s, f_mul(f_mul(is_trans, nl), f_sub(n[KM_PTR_LO], l[KM_PTR_LO])));
  emit(s, f_mul(f_mul(is_trans, nl), f_sub(n[KM_PTR_HI], l[KM_PTR_HI])));
  emit(s, f_mul(l[KM_IS_REAL], f_sub(l[KM_ADDR], f_add(f_add(l[KM_PTR_LO], f_mul(F65536, l[KM_PTR_HI])), f_mul(4, l[KM_IDX])))));
  emit(s, f_mul(l[KM_IS_REAL], f_sub(f_sub(f_add(l[KM_TS], one), l[KM_PTS]), f_add(l[KM_GL], f_mul(F65536, l[KM_GH])))));
}

static void memfinal_constraints(const uint32_t* l, const uint32_t* n, fe is_trans, sink* s) {
  const fe one = 1;
  emit(s, bool_c(l[MF_IS_REAL])); emit(s, bool_c(l[MF_IS_INIT])); emit(s, bool_c(l[MF_BW]));
  emit(s, f_mul(l[MF_IS_INIT], f_sub(one, l[MF_IS_REAL])));
  const fe tn = f_mul(is_trans, n[MF_IS_REAL]);
  emit(s, f_mul(tn, f_sub(one, l[MF_IS_REAL])));
  /* next address - address - 1 = D >= 0, limb by limb with a borrow: every term stays far below p, so this is a
   * statement about integers (all six limbs are looked up in the range table) */
  emit(s, f_mul(tn, f_sub(f_add(f_sub(f_sub(n[MF_LO], l[MF_LO]), one), f_mul(F65536, l[MF_BW])), l[MF_D_LO])));
  emit(s, f_mul(tn, f_sub(f_sub(f_sub(n[MF_HI], l[MF_HI]), l[MF_BW]), l[MF_D_HI])));
  /* an address outside the image is hinted (IS_INIT) or starts as zero (IS_ZERO) */
  const fe z = l[MF_IS_ZERO];
  emit(s, bool_c(z));
  emit(s, f_mul(z, f_sub(one, l[MF_IS_REAL])));
  emit(s, f_mul(z, l[MF_IS_INIT]));
  emit(s, f_mul(z, l[MF_INIT_LO]));
  emit(s, f_mul(z, l[MF_INIT_HI]));
}

/* hint chip (machine.h) */
static void hint_constraints(const uint32_t* l, const uint32_t* n, fe is_first, fe is_last, fe is_trans, sink* s) {
  const fe real = l[HN_IS_REAL], first = l[HN_FIRST], last = l[HN_LAST], used = l[HN_USED];
  emit(s, bool_c(real)); emit(s, bool_c(first)); emit(s, bool_c(last)); emit(s, bool_c(used));
  emit(s, f_mul(f_add(f_add(first, last), used), f_sub(1, real)));
  emit(s, f_mul(f_mul(is_trans, n[HN_IS_REAL]), f_sub(1, real)));
  emit(s, f_mul(is_first, f_sub(real, first)));
  emit(s, f_mul(last, f_sub(l[HN_CNT], 1)));
  const fe go = f_mul(is_trans, f_sub(real, last));
  emit(s, f_mul(go, f_sub(1, n[HN_IS_REAL])));
  emit(s, f_mul(go, f_sub(f_sub(n[HN_ADDR], l[HN_ADDR]), 4)));
  emit(s, f_mul(go, f_add(f_sub(n[HN_CNT], l[HN_CNT]), 1)));
  emit(s, f_mul(is_trans, f_sub(n[HN_FIRST], f_mul(last, n[HN_IS_REAL]))));
  emit(s, f_mul(is_last, f_sub(real, last)));
}

static void mul_constraints(const uint32_t* l, sink* s) {
  emit(s, bool_c(l[MU_IS_REAL])); emit(s, bool_c(l[MU_HI]));
  for (int i = 0; i < 32 + 32 + 64 + 31; ++i) emit(s, bool_c(l[MU_B + i]));
  emit(s, f_mul(l[MU_HI], f_sub(1, l[MU_IS_REAL])));
  fe b[4], c[4], sk[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) { b[i] = byte_of(l, MU_B, i); c[i] = byte_of(l, MU_C, i); }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) sk[i + j] = f_add(sk[i + j], f_mul(b[i], c[j]));
  const fe q0 = bits_val(l, MU_Q0, 10), q1 = bits_val(l, MU_Q1, 11), q2 = bits_val(l, MU_Q2, 10);
  emit(s, f_sub(f_add(sk[0], f_mul(256, sk[1])), f_add(limb_of(l, MU_P, 0), f_mul(F65536, q0))));
  emit(s, f_sub(f_add(f_add(sk[2], f_mul(256, sk[3])), q0), f_add(limb_of(l, MU_P, 1), f_mul(F65536, q1))));
  emit(s, f_sub(f_add(f_add(sk[4], f_mul(256, sk[5])), q1), f_add(limb_of(l, MU_P, 2), f_mul(F65536, q2))));
  emit(s, f_sub(f_add(sk[6], q2), limb_of(l, MU_P, 3)));
  /* mulh (SH) / mulhsu (SHU): the signed high word from the unsigned one */
  const fe sh = l[MU_SH], shu = l[MU_SHU], sg = f_add(sh, shu);
  emit(s, bool_c(sh)); emit(s, bool_c(shu));
  for (int i = 0; i < 4; ++i) emit(s, bool_c(l[MU_K0 + i]));
  emit(s, bool_c(f_add(l[MU_HI], sg))); /* at most one of mulhu, mulh, mulhsu */
  emit(s, f_mul(sg, f_sub(1, l[MU_IS_REAL])));
  {
    const fe b31 = l[MU_B + 31], c31 = l[MU_C + 31], k0 = f_add(l[MU_K0], l[MU_K0 + 1]), k1 = f_add(l[MU_K1], l[MU_K1 + 1]);
    const fe b_lo = limb_of(l, MU_B, 0), b_hi = limb_of(l, MU_B, 1), c_lo = limb_of(l, MU_C, 0), c_hi = limb_of(l, MU_C, 1);
    emit(s, f_add(f_add(f_mul(sg, f_sub(f_sub(l[MU_R], limb_of(l, MU_P, 2)), f_mul(F65536, k0))), f_mul(f_mul(sg, b31), c_lo)),
                  f_mul(f_mul(sh, c31), b_lo)));
    emit(s, f_add(f_add(f_mul(sg, f_sub(f_sub(f_add(l[MU_R + 1], k0), limb_of(l, MU_P, 3)), f_mul(F65536, k1))), f_mul(f_mul(sg, b31), c_hi)),
                  f_mul(f_mul(sh, c31), b_hi)));
  }
}

/* divider chip (machine.h): 52 constraints */
static void div_constraints(const uint32_t* l, sink* s) {
  const fe real = l[DV_IS_REAL];
  emit(s, bool_c(real));
  fe fsum = 0;
  for (int k = 0; k < 4; ++k) { emit(s, bool_c(l[DV_F + k])); fsum = f_add(fsum, l[DV_F + k]); }
  emit(s, f_sub(fsum, real));
  {
    static const int bools[14] = {DV_SN, DV_SD, DV_SQ, DV_SR, DV_CN, DV_CD, DV_CQ, DV_CR, DV_K, DV_BE, DV_NZD, DV_NZQ, DV_NZR, DV_XS};
    for (int k = 0; k < 14; ++k) emit(s, bool_c(l[bools[k]]));
  }
  const fe sgn = f_add(l[DV_F + 0], l[DV_F + 2]); /* div, rem: signed */
  const fe sn = l[DV_SN], sd = l[DV_SD], sq = l[DV_SQ], sr = l[DV_SR];
  emit(s, f_mul(sn, f_sub(1, sgn)));
  emit(s, f_mul(sd, f_sub(1, sgn)));
  emit(s, f_sub(f_sub(l[DV_N + 1], f_mul(32768, sn)), l[DV_NH]));
  emit(s, f_sub(f_sub(l[DV_D + 1], f_mul(32768, sd)), l[DV_DH]));
  {
    static const int xs[4] = {DV_N, DV_D, DV_Q, DV_R}, ax[4] = {DV_AN, DV_AD, DV_AQ, DV_AR}, cx[4] = {DV_CN, DV_CD, DV_CQ, DV_CR};
    const fe sx[4] = {sn, sd, sq, sr};
    for (int k = 0; k < 4; ++k) {
      const fe x_lo = l[xs[k]], x_hi = l[xs[k] + 1], a_lo = l[ax[k]], a_hi = l[ax[k] + 1], c = l[cx[k]];
      emit(s, f_add(f_mul(sx[k], f_sub(f_add(x_lo, a_lo), f_mul(F65536, c))), f_mul(f_sub(1, sx[k]), f_sub(x_lo, a_lo))));
      emit(s, f_add(f_mul(sx[k], f_sub(f_add(f_add(x_hi, a_hi), c), F65536)), f_mul(f_sub(1, sx[k]), f_sub(x_hi, a_hi))));
    }
  }
  emit(s, f_sub(l[DV_XS], f_sub(f_add(sn, sd), f_mul(2, f_mul(sn, sd)))));
  const fe nzd = l[DV_NZD], nzq = l[DV_NZQ], nzr = l[DV_NZR];
  {
    const fe dsum = f_add(l[DV_D], l[DV_D + 1]), qsum = f_add(l[DV_AQ], l[DV_AQ + 1]), rsum = f_add(l[DV_AR], l[DV_AR + 1]);
    emit(s, f_sub(f_mul(dsum, l[DV_INVD]), nzd));
    emit(s, f_mul(f_sub(1, nzd), dsum));
    emit(s, f_mul(nzd, f_sub(1, real)));
    emit(s, f_sub(f_mul(qsum, l[DV_INVQ]), nzq));
    emit(s, f_mul(f_sub(1, nzq), qsum));
    emit(s, f_sub(f_mul(rsum, l[DV_INVR]), nzr));
    emit(s, f_mul(f_sub(1, nzr), rsum));
  }
  emit(s, f_mul(nzd, f_sub(f_sub(f_add(l[DV_PL], l[DV_AR]), l[DV_AN]), f_mul(F65536, l[DV_K]))));
  emit(s, f_mul(nzd, f_sub(f_add(f_add(l[DV_PL + 1], l[DV_AR + 1]), l[DV_K]), l[DV_AN + 1])));
  emit(s, f_mul(nzd, f_sub(f_add(f_sub(f_sub(l[DV_AD], l[DV_AR]), 1), f_mul(F65536, l[DV_BE])), l[DV_E])));
  emit(s, f_mul(nzd, f_sub(f_sub(f_sub(l[DV_AD + 1], l[DV_AR + 1]), l[DV_BE]), l[DV_E + 1])));
  emit(s, f_mul(nzd, f_sub(sq, f_mul(l[DV_XS], nzq))));
  emit(s, f_mul(nzd, f_sub(sr, f_mul(sn, nzr))));
  const fe zd = f_sub(real, nzd);
  emit(s, f_mul(zd, f_sub(l[DV_Q], 65535)));
  emit(s, f_mul(zd, f_sub(l[DV_Q + 1], 65535)));
  emit(s, f_mul(zd, f_sub(l[DV_R], l[DV_N])));
  emit(s, f_mul(zd, f_sub(l[DV_R + 1], l[DV_N + 1])));
  {
    const fe wq = f_add(l[DV_F + 0], l[DV_F + 1]), wr = f_add(l[DV_F + 2], l[DV_F + 3]);
    emit(s, f_sub(f_sub(l[DV_A], f_mul(wq, l[DV_Q])), f_mul(wr, l[DV_R])));
    emit(s, f_sub(f_sub(l[DV_A + 1], f_mul(wq, l[DV_Q + 1])), f_mul(wr, l[DV_R + 1])));
  }
}

/* ALU chip: operands as bits; X is the one-hot shift amount or the comparison difference */
static void alu_constraints(const uint32_t* l, sink* s) {
  const fe one = 1;
#define OPF(op) l[AL_SEL + (op) - OP_SLL]
  emit(s, bool_c(l[AL_IS_REAL]));
  fe selsum = 0;
  for (int k = 0; k < 4; ++k) { emit(s, bool_c(l[AL_SEL + k])); selsum = f_add(selsum, l[AL_SEL + k]); }
  for (int i = 0; i < 96; ++i) emit(s, bool_c(l[AL_B + i])); /* B, C, X */
  emit(s, bool_c(l[AL_K0])); emit(s, bool_c(l[AL_K1]));
  emit(s, f_sub(selsum, l[AL_IS_REAL]));
  const fe a_lo = l[AL_A], a_hi = l[AL_A + 1], b_lo = limb_of(l, AL_B, 0), b_hi = limb_of(l, AL_B, 1);
  const fe c_lo = limb_of(l, AL_C, 0), c_hi = limb_of(l, AL_C, 1), x_lo = limb_of(l, AL_X, 0), x_hi = limb_of(l, AL_X, 1);
  /* ---- shifts: X is the one-hot of the amount ---- */
  {
    const fe sh = f_add(f_add(OPF(OP_SLL), OPF(OP_SRL)), OPF(OP_SRA));
    fe sum = 0, idx = 0;
    for (int k = 0; k < 32; ++k) { sum = f_add(sum, l[AL_X + k]); idx = f_add(idx, f_mul((fe)k, l[AL_X + k])); }
    emit(s, f_mul(sh, f_sub(sum, one)));
    emit(s, f_mul(sh, f_sub(idx, bits_val(l, AL_C, 5))));
    for (int kind = 0; kind < 3; ++kind) {
      fe t[32];
      for (int j = 0; j < 32; ++j) {
        fe acc = 0;
        for (int k = 0; k < 32; ++k) {
          int src;
          if (kind == 0) { if (k > j) continue; src = j - k; }
          else if (kind == 1) { if (j + k > 31) continue; src = j + k; }
          else src = j + k > 31 ? 31 : j + k;
          acc = f_add(acc, f_mul(l[AL_X + k], l[AL_B + src]));
        }
        t[j] = acc;
      }
      const fe sel = OPF(kind == 0 ? OP_SLL : kind == 1 ? OP_SRL : OP_SRA);
      for (int h = 0; h < 2; ++h) {
        fe acc = 0;
        for (int i = 15; i >= 0; --i) acc = f_add(f_add(acc, acc), t[16 * h + i]);
        emit(s, f_mul(sel, f_sub(h ? a_hi : a_lo, acc)));
      }
    }
  }
  /* ---- signed less-than: X = B - C (mod 2^32) with the sign bits swapped, K1 = "less than" ---- */
  {
    const fe cmp = OPF(OP_SLT);
    emit(s, f_mul(cmp, f_sub(f_add(f_sub(b_lo, c_lo), f_mul(F65536, l[AL_K0])), x_lo)));
    emit(s, f_mul(cmp, f_add(f_sub(f_add(f_sub(f_sub(b_hi, c_hi), l[AL_K0]), f_mul(F65536, l[AL_K1])), x_hi),
                             f_mul(F65536, f_sub(l[AL_C + 31], l[AL_B + 31])))));
    emit(s, f_mul(cmp, f_sub(a_lo, l[AL_K1])));
    emit(s, f_mul(cmp, a_hi));
  }
#undef OPF
}

/* bitwise chip: one operation per real row; the byte lookups do the rest */
static void bw_constraints(const uint32_t* l, sink* s) {
  emit(s, bool_c(l[BW_IS_REAL]));
  fe selsum = 0;
  for (int k = 0; k < 3; ++k) { emit(s, bool_c(l[BW_SEL + k])); selsum = f_add(selsum, l[BW_SEL + k]); }
  emit(s, f_sub(selsum, l[BW_IS_REAL]));
}

/* The permutation of one row: the columns IN (16 words), EXT (8 rounds x (16 cubes, 16 seventh powers)), INT (13 x (cube,
 * seventh power)); 282 constraints; st[] comes back as the 16 output words, linear in the last round's columns. */
static void p2_perm_constraints(const uint32_t* l, int c_in, int c_ext, int c_int, fe st[16], sink* s) {
  uint32_t ext_rc[8][16], int_rc[13];
  orc_poseidon2_constants(&ext_rc[0][0], int_rc);
  for (int i = 0; i < 16; ++i) st[i] = l[c_in + i];
  orc_p2_external_linear(st);
  for (int rd = 0; rd < 8; ++rd) {
    if (rd == 4)
      for (int ir = 0; ir < 13; ++ir) {
        const fe x = f_add(st[0], int_rc[ir]), x3 = l[c_int + 2 * ir], y = l[c_int + 2 * ir + 1];
        emit(s, f_sub(x3, f_mul(f_mul(x, x), x)));
        emit(s, f_sub(y, f_mul(f_mul(x3, x3), x)));
        st[0] = y;
        orc_p2_internal_linear(st);
      }
    for (int i = 0; i < 16; ++i) {
      const fe x = f_add(st[i], ext_rc[rd][i]), x3 = l[c_ext + 32 * rd + i], y = l[c_ext + 32 * rd + 16 + i];
      emit(s, f_sub(x3, f_mul(f_mul(x, x), x)));
      emit(s, f_sub(y, f_mul(f_mul(x3, x3), x)));
      st[i] = y;
    }
    orc_p2_external_linear(st);
  }
}
static inline fe4 row4(const uint32_t* row, int col) { fe4 r; for (int i = 0; i < 4; ++i) r.c[i] = row[col + i]; return r; }
static inline void emit4(sink* s, fe4 v) { for (int i = 0; i < 4; ++i) emit(s, v.c[i]); }
static inline void emit4_sel(sink* s, fe sel, fe4 v) { for (int i = 0; i < 4; ++i) emit(s, f_mul(sel, v.c[i])); }

/* Poseidon2 chip: every S-box through its cube; the state between S-boxes is linear in the columns.  Then the 67
 * constraints that tie the rows of an opening together (machine.h "Poseidon2 chip"); they refer to the next row wherever
 * that row takes over from this one, cyclically: row 0 takes over from nothing.  Then format v16's 45: the ends of runs and of
 * hashes, and Horner's rule over the absorbed words. */
static void p2_constraints(const uint32_t* l, const uint32_t* n, fe is_first, fe is_trans, sink* s) {
  const fe real = l[P2_IS_REAL];
  emit(s, bool_c(real));
  emit(s, f_mul(f_mul(is_trans, n[P2_IS_REAL]), f_sub(1, real))); /* the real rows are a prefix */
  fe st[16];
  p2_perm_constraints(l, P2_IN, P2_EXT, P2_INT, st, s);
  /* st[] = the permutation's 16 output words.  Row kinds: one per real row; NEW lives on sponge rows, FR on first blocks,
   * SND and SE on sponge rows, RE on path and injection rows */
  const fe fn = l[P2_FN], sz = l[P2_SZ], sc = l[P2_SC], pl = l[P2_PL], pr = l[P2_PR], fj = l[P2_FJ], nw = l[P2_NEW], snd = l[P2_SND],
           fr = l[P2_FR];
  emit(s, bool_c(fn)); emit(s, bool_c(sz)); emit(s, bool_c(sc)); emit(s, bool_c(pl)); emit(s, bool_c(pr));
  emit(s, bool_c(fj)); emit(s, bool_c(nw)); emit(s, bool_c(snd)); emit(s, bool_c(fr));
  const fe chain = f_add(f_add(sc, pl), f_add(pr, fj)); /* the kinds that take over from the row before */
  emit(s, f_sub(f_add(f_add(fn, sz), chain), real));
  emit(s, f_mul(nw, f_sub(f_sub(1, sz), sc)));
  emit(s, f_mul(snd, f_sub(f_sub(1, sz), sc)));
  emit(s, f_mul(fr, f_sub(1, sz)));
  /* a first block starts from the zero state; the first block of a run from K = 1, M = 0 */
  for (int i = 0; i < 8; ++i) emit(s, f_mul(sz, l[P2_IN + 8 + i]));
  const fe key = f_add(l[P2_KL], f_mul(F65536, l[P2_KH])), m = l[P2_M];
  emit(s, f_mul(f_mul(sz, nw), f_sub(key, 1)));
  emit(s, f_mul(f_mul(sz, nw), m));
  emit(s, f_mul(is_first, chain)); /* nothing precedes row 0 */
  /* the next row, where it takes over from this one */
  const fe nsc = n[P2_SC], npl = n[P2_PL], npr = n[P2_PR], nfj = n[P2_FJ], npath = f_add(npl, npr);
  const fe nkey = f_add(n[P2_KL], f_mul(F65536, n[P2_KH])), nm = n[P2_M];
  emit(s, f_mul(f_add(f_add(nsc, npath), nfj), f_sub(n[P2_T], l[P2_T])));
  /* ... a sponge goes on: the capacity, the labels and the flag of its run; only a sponge row precedes it */
  for (int i = 0; i < 8; ++i) emit(s, f_mul(nsc, f_sub(n[P2_IN + 8 + i], st[8 + i])));
  emit(s, f_mul(nsc, f_sub(nkey, key)));
  emit(s, f_mul(nsc, f_sub(nm, m)));
  emit(s, f_mul(nsc, f_sub(n[P2_NEW], nw)));
  emit(s, f_mul(nsc, f_sub(f_sub(1, sz), sc)));
  /* ... a path step: the running digest on its side, one more position bit, one more level; it follows the leaf's sponge
   * (a run's, not an injected matrix's), a path step or an injection */
  for (int i = 0; i < 8; ++i) emit(s, f_mul(npl, f_sub(n[P2_IN + i], st[i])));
  for (int i = 0; i < 8; ++i) emit(s, f_mul(npr, f_sub(n[P2_IN + 8 + i], st[i])));
  emit(s, f_mul(npath, f_sub(f_sub(nkey, f_add(key, key)), npr)));
  emit(s, f_mul(npath, f_sub(nm, f_add(m, m))));
  emit(s, f_mul(npath, f_sub(f_sub(f_sub(f_sub(1, f_mul(f_add(sz, sc), nw)), pl), pr), fj)));
  /* ... an injection: the running digest on the left, the level marked; only a path step precedes it */
  for (int i = 0; i < 8; ++i) emit(s, f_mul(nfj, f_sub(n[P2_IN + i], st[i])));
  emit(s, f_mul(nfj, f_sub(nkey, key)));
  emit(s, f_mul(nfj, f_sub(f_sub(nm, m), 1)));
  emit(s, f_mul(nfj, f_sub(f_sub(1, pl), pr)));
  /* ---- format v16 (stage 2b) ---- */
  const fe re = l[P2_RE], se = l[P2_SE];
  emit(s, bool_c(re));
  emit(s, bool_c(se));
  emit(s, f_mul(re, f_sub(f_sub(f_sub(1, pl), pr), fj)));
  emit(s, f_mul(se, f_sub(f_sub(1, sz), sc)));
  emit(s, f_mul(se, nsc));
  fe4 ap[8];
  for (int j = 0; j < 8; ++j) ap[j] = row4(l, P2_AP + 4 * j);
  for (int j = 0; j < 7; ++j) emit4(s, e_sub(ap[j + 1], e_mul(ap[j], ap[0])));
  for (int i = 0; i < 4; ++i) emit(s, f_mul(nsc, f_sub(n[P2_AP + i], ap[0].c[i])));
  {
    fe4 bv = e_zero(); /* this row's block: sum_{i < 8} alpha^(7 - i) in_i */
    for (int i = 0; i < 7; ++i) bv = e_add(bv, e_mul_base(ap[6 - i], l[P2_IN + i]));
    bv.c[0] = f_add(bv.c[0], l[P2_IN + 7]);
    emit4_sel(s, sz, e_sub(row4(l, P2_SO), bv));
    fe4 nbv = e_zero();
    for (int i = 0; i < 7; ++i) nbv = e_add(nbv, e_mul_base(row4(n, P2_AP + 4 * (6 - i)), n[P2_IN + i]));
    nbv.c[0] = f_add(nbv.c[0], n[P2_IN + 7]);
    emit4_sel(s, nsc, e_sub(e_sub(row4(n, P2_SO), e_mul(row4(l, P2_SO), row4(n, P2_AP + 28))), nbv));
  }
}

/* Query chip (machine.h): the constraints in the order of zk-state-proofs_amd/csrc/device/air_machine.hpp eval_qr */
static void qr_constraints(const uint32_t* l, const uint32_t* n, fe is_first, fe is_trans, sink* s) {
  const fe real = l[QR_IS_REAL], first = l[QR_FIRST], last = l[QR_LAST], bit = l[QR_BIT], eq = l[QR_EQ], f1 = l[QR_F1], f2 = l[QR_F2],
           f3 = l[QR_F3], csr = l[QR_CSR], fl = l[QR_FL], lay = l[QR_LAY], cs = l[QR_CS], pr0 = l[QR_PR0], p0a = l[QR_P0A],
           hasro = l[QR_HASRO], has0 = l[QR_HAS0];
  {
    const fe bools[17] = {real, first, last, bit, eq, f1, f2, f3, csr, fl, lay, cs, pr0, p0a, hasro, has0, l[QR_CNT0]};
    for (int i = 0; i < 17; ++i) emit(s, bool_c(bools[i]));
  }
  const fe cont = f_sub(n[QR_IS_REAL], n[QR_FIRST]), nl1 = f_sub(n[QR_LAY], n[QR_FL]);
#define ADD(a, b) f_add(a, b)
#define SUB(a, b) f_sub(a, b)
#define MUL(a, b) f_mul(a, b)
#define DBL(a) f_add(a, a)
  emit(s, MUL(MUL(is_trans, n[QR_IS_REAL]), SUB(1, real)));
  emit(s, MUL(is_first, SUB(real, first)));
  {
    fe sum = first;
    const fe fl_[14] = {last, csr, fl, lay, pr0, p0a, hasro, has0, bit, eq, f1, f2, f3, cs};
    for (int i = 0; i < 14; ++i) sum = ADD(sum, fl_[i]);
    emit(s, MUL(sum, SUB(1, real)));
  }
  const fe j = l[QR_J];
  emit(s, MUL(first, SUB(j, 30)));
  emit(s, MUL(last, j));
  emit(s, MUL(cont, ADD(SUB(n[QR_J], j), 1)));
  emit(s, MUL(last, cont));
  emit(s, MUL(SUB(real, last), SUB(1, cont)));
  emit(s, MUL(cont, SUB(n[QR_LEAF], l[QR_LEAF])));
  emit(s, MUL(cont, SUB(n[QR_QL], l[QR_QL])));
  emit(s, MUL(first, SUB(l[QR_ACC], bit)));
  emit(s, MUL(cont, SUB(SUB(n[QR_ACC], DBL(l[QR_ACC])), n[QR_BIT])));
  emit(s, MUL(first, ADD(ADD(f1, f2), f3)));
  emit(s, MUL(cont, SUB(n[QR_F1], first)));
  emit(s, MUL(cont, SUB(n[QR_F2], f1)));
  emit(s, MUL(cont, SUB(n[QR_F3], f2)));
  emit(s, MUL(first, SUB(eq, bit)));
  {
    const fe nf = ADD(ADD(n[QR_F1], n[QR_F2]), n[QR_F3]);
    emit(s, MUL(nf, SUB(n[QR_EQ], MUL(eq, n[QR_BIT]))));
    emit(s, MUL(SUB(cont, nf), SUB(n[QR_EQ], eq)));
    emit(s, MUL(MUL(SUB(SUB(SUB(SUB(real, first), f1), f2), f3), eq), bit));
  }
  emit(s, MUL(csr, lay));
  emit(s, MUL(fl, SUB(1, lay)));
  emit(s, MUL(first, ADD(csr, lay)));
  emit(s, MUL(cont, SUB(n[QR_FL], csr)));
  emit(s, MUL(cont, SUB(SUB(n[QR_LAY], csr), lay)));
  emit(s, MUL(last, SUB(1, lay)));
  emit(s, MUL(fl, l[QR_K]));
  emit(s, MUL(nl1, SUB(SUB(n[QR_K], l[QR_K]), 1)));
  emit(s, MUL(csr, SUB(cs, bit)));
  emit(s, MUL(MUL(cont, ADD(csr, lay)), SUB(n[QR_CS], cs)));
  emit(s, MUL(last, SUB(l[QR_POW], 1)));
  emit(s, MUL(last, l[QR_LOW]));
  emit(s, MUL(last, l[QR_REV]));
  emit(s, MUL(cont, SUB(l[QR_POW], DBL(n[QR_POW]))));
  emit(s, MUL(cont, SUB(SUB(l[QR_LOW], n[QR_LOW]), MUL(n[QR_BIT], n[QR_POW]))));
  emit(s, MUL(cont, SUB(SUB(l[QR_REV], DBL(n[QR_REV])), n[QR_BIT])));
  emit(s, MUL(pr0, SUB(j, 16)));
  emit(s, MUL(first, SUB(l[QR_CNT0], pr0)));
  emit(s, MUL(cont, SUB(SUB(n[QR_CNT0], l[QR_CNT0]), n[QR_PR0])));
  emit(s, MUL(last, SUB(l[QR_CNT0], 1)));
  emit(s, MUL(first, p0a));
  emit(s, MUL(cont, SUB(SUB(n[QR_P0A], p0a), pr0)));
  emit(s, MUL(MUL(cont, pr0), SUB(n[QR_KEY0], 1)));
  emit(s, MUL(MUL(cont, pr0), n[QR_M0]));
  emit(s, MUL(MUL(cont, p0a), SUB(SUB(n[QR_KEY0], DBL(l[QR_KEY0])), bit)));
  emit(s, MUL(MUL(cont, p0a), SUB(SUB(n[QR_M0], DBL(l[QR_M0])), n[QR_HAS0])));
  emit(s, MUL(has0, SUB(1, p0a)));
  emit(s, MUL(has0, SUB(1, hasro)));
  emit(s, MUL(cont, SUB(n[QR_MT0], l[QR_MT0])));
  emit(s, MUL(last, SUB(l[QR_MT0], MUL(4, l[QR_M0]))));
  emit(s, MUL(fl, SUB(l[QR_KEYJ], 1)));
  emit(s, MUL(fl, l[QR_MJ]));
  emit(s, MUL(nl1, SUB(SUB(n[QR_KEYJ], DBL(l[QR_KEYJ])), bit)));
  emit(s, MUL(nl1, SUB(SUB(n[QR_MJ], DBL(l[QR_MJ])), n[QR_HASRO])));
  emit(s, MUL(cont, SUB(n[QR_MT], l[QR_MT])));
  emit(s, MUL(last, SUB(l[QR_MT], MUL(4, l[QR_MJ]))));
  emit(s, MUL(hasro, SUB(1, lay)));
  emit(s, MUL(fl, SUB(1, hasro)));
  const fe omi = l[QR_OMI];
  emit(s, MUL(cont, SUB(n[QR_OMI], omi)));
  emit(s, SUB(SUB(l[QR_MU], real), MUL(bit, SUB(omi, 1))));
  emit(s, SUB(SUB(l[QR_CSM], real), MUL(cs, SUB(omi, 1))));
  emit(s, SUB(l[QR_R2], MUL(l[QR_R], l[QR_R])));
  emit(s, MUL(fl, SUB(l[QR_R], l[QR_MU])));
  emit(s, MUL(nl1, SUB(n[QR_R], MUL(l[QR_R2], n[QR_MU]))));
  emit(s, MUL(cont, SUB(n[QR_YT], l[QR_YT])));
  emit(s, MUL(last, SUB(l[QR_YT], MUL(l[QR_R2], l[QR_CSM]))));
  emit(s, MUL(fl, SUB(l[QR_YKI], l[QR_YT])));
  emit(s, MUL(nl1, SUB(n[QR_YKI], MUL(l[QR_YKI], l[QR_YKI]))));
  emit(s, MUL(fl, SUB(l[QR_GI], F_GEN_INV)));
  emit(s, MUL(nl1, SUB(n[QR_GI], MUL(l[QR_GI], l[QR_GI]))));
  emit(s, SUB(l[QR_XINV], MUL(MUL(l[QR_GI], l[QR_YKI]), SUB(1, DBL(bit)))));
  /* the fold */
  const fe4 lo = row4(l, QR_LO), hi = row4(l, QR_HI), be = row4(l, QR_BETA), d = e_sub(lo, hi);
  for (int i = 0; i < 4; ++i) emit(s, ADD(SUB(l[QR_E + i], lo.c[i]), MUL(bit, d.c[i])));
  {
    const fe4 pd = e_mul(be, d);
    for (int i = 0; i < 4; ++i) emit(s, SUB(SUB(SUB(DBL(l[QR_F + i]), lo.c[i]), hi.c[i]), MUL(l[QR_XINV], pd.c[i])));
  }
  for (int i = 0; i < 4; ++i) emit(s, MUL(SUB(1, hasro), l[QR_RO + i]));
  for (int i = 0; i < 4; ++i) emit(s, MUL(fl, SUB(l[QR_E + i], l[QR_RO + i])));
  for (int i = 0; i < 4; ++i) emit(s, MUL(nl1, SUB(SUB(n[QR_E + i], l[QR_F + i]), n[QR_RO + i])));
  /* the reduced opening */
  {
    const fe4 dl = row4(l, QR_DL), d2 = row4(l, QR_D2), d3 = row4(l, QR_D3), d4 = row4(l, QR_D4), g2 = row4(l, QR_G2),
              zeta = row4(l, QR_ZETA), zw = row4(l, QR_ZW), d0 = row4(l, QR_D0), d1 = row4(l, QR_D1), h0 = row4(l, QR_H),
              h1 = row4(l, QR_H + 4), h2 = row4(l, QR_H + 8), h3 = row4(l, QR_H + 12);
    emit4(s, e_sub(d2, e_mul(dl, dl)));
    emit4(s, e_sub(d3, e_mul(d2, dl)));
    emit4(s, e_sub(d4, e_mul(d2, d2)));
    emit4(s, e_sub(g2, e_add(h1, e_mul(dl, h2))));
    emit4(s, e_sub(zw, e_mul_base(zeta, l[QR_WH])));
    const fe yki = l[QR_YKI], nyki = SUB(0, yki);
    fe4 den0 = e_mul_base(zeta, nyki), den1 = e_mul_base(zw, nyki);
    den0.c[0] = ADD(den0.c[0], F_GEN);
    den1.c[0] = ADD(den1.c[0], F_GEN);
    fe4 p0 = e_mul(d0, den0), p1 = e_mul(d1, den1);
    p0.c[0] = SUB(p0.c[0], yki);
    p1.c[0] = SUB(p1.c[0], yki);
    emit4(s, p0);
    emit4(s, p1);
    fe4 hs = e_add(e_add(h0, e_mul(dl, h1)), e_add(e_mul(d2, h2), e_mul(d3, h3)));
    hs = e_sub(hs, row4(l, QR_B1));
    const fe4 t2 = e_sub(e_mul(d4, g2), row4(l, QR_B2));
    emit4(s, e_sub(row4(l, QR_RO), e_add(e_mul(d0, hs), e_mul(d1, t2))));
  }
#undef ADD
#undef SUB
#undef MUL
#undef DBL
}

/* Transcript chip (machine.h) */
static void tr_constraints(const uint32_t* l, const uint32_t* n, fe is_first, fe is_trans, sink* s) {
  const fe real = l[TR_IS_REAL], first = l[TR_FIRST], abs_ = l[TR_ABS];
  emit(s, bool_c(real));
  emit(s, f_mul(f_mul(is_trans, n[TR_IS_REAL]), f_sub(1, real)));
  fe st[16];
  p2_perm_constraints(l, TR_IN, TR_EXT, TR_INT, st, s);
  fe fsum = f_add(first, abs_);
  emit(s, bool_c(first));
  emit(s, bool_c(abs_));
  for (int k = 0; k < 7 + 8; ++k) {
    emit(s, bool_c(l[TR_UROOT + k]));
    fsum = f_add(fsum, l[TR_UROOT + k]);
  }
  emit(s, f_mul(fsum, f_sub(1, real)));
  emit(s, f_mul(first, f_sub(1, abs_)));
  emit(s, f_mul(first, l[TR_STEP]));
  for (int i = 0; i < 8; ++i) emit(s, f_mul(first, l[TR_IN + 8 + i]));
  emit(s, f_mul(is_first, f_sub(real, first)));
  const fe cont = f_sub(n[TR_IS_REAL], n[TR_FIRST]);
  emit(s, f_mul(cont, f_sub(n[TR_LEAF], l[TR_LEAF])));
  emit(s, f_mul(cont, f_sub(f_sub(n[TR_STEP], l[TR_STEP]), 1)));
  for (int i = 0; i < 8; ++i) emit(s, f_mul(cont, f_sub(n[TR_IN + 8 + i], st[8 + i])));
  {
    const fe sq = f_mul(cont, f_sub(1, n[TR_ABS]));
    for (int i = 0; i < 8; ++i) emit(s, f_mul(sq, f_sub(n[TR_IN + i], st[i])));
  }
  {
    fe qm = l[TR_QM];
    for (int k = 1; k < 8; ++k) qm = f_add(qm, l[TR_QM + k]);
    emit(s, f_mul(qm, f_sub(1, l[TR_UQ])));
  }
  emit(s, f_mul(l[TR_MROOT], f_sub(1, l[TR_UROOT])));
  emit(s, f_mul(l[TR_MFIN], f_sub(1, l[TR_UFIN])));
  emit(s, f_mul(l[TR_MZETA], f_sub(1, l[TR_UZETA])));
  emit(s, f_mul(l[TR_MAF], f_sub(1, l[TR_UAF])));
  emit(s, f_mul(l[TR_MBETA], f_sub(1, l[TR_UBETA])));
}

/* sub-word chip: M is the memory word, C the low limb of the stored register, both as bits */
static void sub_constraints(const uint32_t* l, sink* s) {
  enum { LB = 0, LH, LBU, LHU, SB, SH };
#define SF(k) l[SW_SEL + (k)]
  emit(s, bool_c(l[SW_IS_REAL]));
  fe selsum = 0, osum = 0;
  for (int k = 0; k < 6; ++k) { emit(s, bool_c(SF(k))); selsum = f_add(selsum, SF(k)); }
  for (int k = 0; k < 4; ++k) { emit(s, bool_c(l[SW_O + k])); osum = f_add(osum, l[SW_O + k]); }
  emit(s, bool_c(l[SW_S]));
  emit(s, f_sub(selsum, l[SW_IS_REAL]));
  emit(s, f_sub(osum, l[SW_IS_REAL]));
  const fe a_lo = l[SW_A], a_hi = l[SW_A + 1], sgn = l[SW_S], selb = l[SW_SELB];
  const fe o0 = l[SW_O], o1 = l[SW_O + 1], o2 = l[SW_O + 2], o3 = l[SW_O + 3];
  const fe mb[4] = {l[SW_MB], l[SW_MB + 1], l[SW_MB + 2], l[SW_MB + 3]}, cb = l[SW_CB];
  const fe m_lo = f_add(mb[0], f_mul(256, mb[1])), m_hi = f_add(mb[2], f_mul(256, mb[3])), c_lo = f_add(cb, f_mul(256, l[SW_CB + 1]));
  /* half-word accesses are 2-aligned */
  emit(s, f_mul(f_add(f_add(SF(LH), SF(LHU)), SF(SH)), f_add(o1, o3)));
  /* the sign: only signed loads have one, and it is bit 7 (byte-operation lookup) of the accessed byte / of the
   * accessed half-word's upper byte */
  fe bv = 0;
  for (int p = 0; p < 4; ++p) bv = f_add(bv, f_mul(l[SW_O + p], mb[p]));
  const fe hv = f_add(f_mul(o0, m_lo), f_mul(o2, m_hi)), hb = f_add(f_mul(o0, mb[1]), f_mul(o2, mb[3]));
  emit(s, f_mul(f_sub(f_sub(1, SF(LB)), SF(LH)), sgn));
  emit(s, f_mul(SF(LB), f_sub(selb, bv)));
  emit(s, f_mul(SF(LH), f_sub(selb, hb)));
  emit(s, f_mul(SF(LHU), f_sub(a_lo, hv)));
  emit(s, f_mul(SF(LHU), a_hi));
  emit(s, f_mul(SF(LH), f_sub(a_lo, hv)));
  emit(s, f_mul(SF(LH), f_sub(a_hi, f_mul(65535, sgn))));
  emit(s, f_mul(SF(LBU), f_sub(a_lo, bv)));
  emit(s, f_mul(SF(LBU), a_hi));
  emit(s, f_mul(SF(LB), f_sub(a_lo, f_add(bv, f_mul(0xff00, sgn)))));
  emit(s, f_mul(SF(LB), f_sub(a_hi, f_mul(65535, sgn))));
  /* stores: A is the word left behind - the old word with the low half-word / byte of the stored register put in */
  emit(s, f_mul(SF(SH), f_sub(f_sub(a_lo, m_lo), f_mul(o0, f_sub(c_lo, m_lo)))));
  emit(s, f_mul(SF(SH), f_sub(f_sub(a_hi, m_hi), f_mul(o2, f_sub(c_lo, m_hi)))));
  emit(s, f_mul(SF(SB), f_sub(f_sub(a_lo, m_lo), f_add(f_mul(o0, f_sub(cb, mb[0])), f_mul(256, f_mul(o1, f_sub(cb, mb[1])))))));
  emit(s, f_mul(SF(SB), f_sub(f_sub(a_hi, m_hi), f_add(f_mul(o2, f_sub(cb, mb[2])), f_mul(256, f_mul(o3, f_sub(cb, mb[3])))))));
  /* loads put no register limb on the bus */
  emit(s, f_mul(f_add(f_add(SF(LB), SF(LH)), f_add(SF(LBU), SF(LHU))), c_lo));
#undef SF
}

static void ecall_constraints(const uint32_t* l, const uint32_t* pub, sink* s) {
  static const uint32_t codes[6] = {0x00, 0x02, 0x10, 0x1a, 0xf0, 0xf1};
  const fe real = l[EC_IS_REAL];
  emit(s, bool_c(real));
  fe scsum = 0, code = 0;
  for (int k = 0; k < 6; ++k) {
    emit(s, bool_c(l[EC_SC + k]));
    scsum = f_add(scsum, l[EC_SC + k]);
    code = f_add(code, f_mul(codes[k], l[EC_SC + k]));
  }
  emit(s, f_sub(scsum, real));
  /* t0 holds one of the six codes and is rewritten with itself, except by HINT_LEN (whose answer the CPU row range-checks) */
  emit(s, f_sub(l[EC_B_LO], code));
  const fe same = f_sub(real, l[EC_SC + SC_HINT_LEN]);
  emit(s, f_mul(same, f_sub(l[EC_A_LO], l[EC_B_LO])));
  emit(s, f_mul(same, l[EC_A_HI]));
  /* the next instruction, except that HALT goes to the padding instruction */
  const fe pc4 = f_add(l[EC_PC], 4);
  emit(s, f_sub(f_mul(real, f_sub(l[EC_NP], pc4)), f_mul(l[EC_SC + SC_HALT], f_sub(pub[CPUPUB_PAD_PC] % FP, pc4))));
  /* COMMIT / COMMIT_DEFERRED: the word index in a0 is the whole register (the PUBC tuple carries its low limb only) */
  emit(s, f_mul(f_add(l[EC_SC + SC_COMMIT], l[EC_SC + SC_DEFER]), l[EC_C_HI]));
  /* HINT_READ: the number of words the read covers (NW is looked up as 16 bits) */
  const fe hr = l[EC_SC + SC_HINT_READ], p1 = l[EC_P1], p2 = l[EC_P2];
  emit(s, bool_c(p1));
  emit(s, bool_c(p2));
  emit(s, f_mul(f_add(p1, p2), f_sub(1, hr)));
  emit(s, f_mul(f_sub(1, hr), l[EC_NW]));
  emit(s, f_mul(hr, f_sub(f_sub(f_sub(f_sub(f_mul(4, l[EC_NW]), p1), f_add(p2, p2)), l[EC_M_LO]), f_mul(F65536, l[EC_M_HI]))));
}

static void run_constraints(int chip, const uint32_t* prep, const uint32_t* loc, const uint32_t* nxt, uint32_t is_first,
                            uint32_t is_last, uint32_t is_trans, const uint32_t* pub, sink* s) {
  static const uint32_t no_pub[CPUPUB_N] = {0, 0, 0, 0, 0};
  switch (chip) {
    case CH_CPU: case CH_CPU2: case CH_CPU3: case CH_CPU4: case CH_CPU5: case CH_CPU6: case CH_CPU7: case CH_CPU8:
      cpu_constraints(loc, nxt, is_first, is_last, is_trans, pub ? pub : no_pub, s); break;
    case CH_KECCAK:
      if (s->out) orc_keccak_constraints(loc, nxt, is_first, is_last, is_trans, s->out + s->k);
      s->k += KA_NUM_CONSTRAINTS;
      /* the call time is constant inside a permutation's 24 rows */
      emit(s, f_mul(f_mul(is_trans, f_sub(1, loc[KA_FLAGS + 23])), f_sub(nxt[KC_TS], loc[KC_TS])));
      break;
    case CH_KMEM: kmem_constraints(loc, nxt, is_first, is_trans, s); break;
    case CH_MEMFINAL: memfinal_constraints(loc, nxt, is_trans, s); break;
    case CH_IMAGE: emit(s, f_sub(loc[0], prep[IMG_P_REAL])); break; /* every image word is sent exactly once */
    case CH_PROGRAM: break;
    case CH_MUL: mul_constraints(loc, s); break;
    case CH_BW:
    case CH_BW2: bw_constraints(loc, s); break;
    case CH_P2: p2_constraints(loc, nxt, is_first, is_trans, s); break;
    case CH_QR: qr_constraints(loc, nxt, is_first, is_trans, s); break;
    case CH_TR: tr_constraints(loc, nxt, is_first, is_trans, s); break;
    case CH_DIV: div_constraints(loc, s); break;
    case CH_HINT: hint_constraints(loc, nxt, is_first, is_last, is_trans, s); break;
    case CH_TABLE: /* only multiples of 4 answer aligned lookups, only values 1 .. ADDR_HI_MAX high-address-limb lookups */
      emit(s, f_mul(loc[TB_M_AL], prep[TB_P_NA]));
      emit(s, f_mul(loc[TB_M_TOP], prep[TB_P_NT]));
      break;
    case CH_ALU:
    case CH_ALU2: alu_constraints(loc, s); break;
    case CH_SUB:
    case CH_SUB2: sub_constraints(loc, s); break;
    case CH_ECALL: ecall_constraints(loc, pub ? pub : no_pub, s); break;
  }
}

static int count_constraints(int chip) {
  sink s = {NULL, 0};
  static uint32_t zeros[KECCAK_WIDTH + 8];
  run_constraints(chip, zeros, zeros, zeros, 0, 0, 0, NULL, &s);
  return s.k;
}

void orc_machine_constraints(int chip, const uint32_t* prep, const uint32_t* loc, const uint32_t* nxt, uint32_t is_first,
                             uint32_t is_last, uint32_t is_trans, const uint32_t* pub, uint32_t* out) {
  build();
  sink s = {out, 0};
  run_constraints(chip, prep, loc, nxt, is_first, is_last, is_trans, pub, &s);
}
