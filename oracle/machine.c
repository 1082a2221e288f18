/* ORACLE -- TEST INFRASTRUCTURE ONLY (see field.h and machine.h).
 *
 * Chips of the machine proof: bus interactions, trace generation from the executor's records and
 * base-field constraints.  This repository's own arithmetisation (machine.h header note); what it
 * must reproduce is the reference's statement: the committed RV32IM guest
 * (circuits/sp1-merkle-proof/src/main.rs:4-14 running crypto-ops/src/lib.rs:8-23) executed from its
 * entry point to HALT with the committed public values.  PARITY UNPINNED vs sp1-core-machine 3.4.0
 * (reference Cargo.lock:7130; sources absent).
 *
 * Design in one paragraph.  One CPU row per cycle; operands live as BITS (A = value written, B =
 * reg[rs1], C = reg[rs2] or the immediate, M = memory word read), so bitwise operations, shifts,
 * comparisons, byte/half selection and every range check are polynomial identities of degree <= 3
 * inside the row and no byte-lookup or ALU tables are needed.  Registers are memory addresses
 * 0..31.  Memory consistency is the offline argument: every access consumes the tuple
 * (addr, value, time) its predecessor produced and produces its own; Image (preprocessed program
 * image + zeroed registers) and MemFinal (every touched address once, strictly increasing) open
 * and close each address; "previous time < time": the difference is two 12-bit limbs, each looked
 * up in the Range table (a preprocessed column 0..4095 with a multiplicity column).  Instruction fetch is
 * a lookup into the preprocessed Program table.  The keccak precompile call hands (time, pointer)
 * to KeccakMem, which moves the 50 state words through the memory bus and matches them, word by
 * word, against what the keccak-f chip exports.  mul / mulhu go to a multiplier chip.  COMMIT and
 * HALT post the public digest words and the exit code on a bus the verifier closes.
 */
#include <stdlib.h>
#include <string.h>

#include "machine.h"

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
/* 16-bit limb of a little-endian bit block starting at column `bits` */
static orc_lf lf_limb(int bits, int limb) {
  orc_lf f;
  lf_zero(&f);
  for (int i = 0; i < 16; ++i) lf_add(&f, bits + 16 * limb + i, 1u << i);
  return f;
}
static orc_lf lf_plus(orc_lf f, uint32_t c) { f.c0 = f_add(f.c0, c % FP); return f; }

static orc_inter g_cpu[21], g_keccak[50], g_kmem[6], g_memfinal[2], g_image[1], g_program[1], g_mul[2], g_range[1];
static orc_chip g_chips[N_CHIPS];
static int g_ready = 0;

static orc_inter mem_inter(int sign, orc_lf mult, orc_lf addr, orc_lf lo, orc_lf hi, orc_lf ts) {
  orc_inter it;
  it.bus = BUS_MEM; it.sign = sign; it.mult = mult; it.n_el = 4;
  it.el[0] = addr; it.el[1] = lo; it.el[2] = hi; it.el[3] = ts;
  return it;
}

static orc_inter range_inter(int sign, orc_lf mult, orc_lf value) {
  orc_inter it;
  it.bus = BUS_RANGE; it.sign = sign; it.mult = mult; it.n_el = 1;
  it.el[0] = value;
  return it;
}

static int count_constraints(int chip);

static void build(void) {
  if (g_ready) return;
  /* in the CPU chip there are no preprocessed columns: linear forms index main columns directly */
  orc_lf is_real = lf_col(C_IS_REAL), ts = lf_col(C_TS);
  orc_lf a_lo = lf_col(C_A), a_hi = lf_col(C_A + 1), b_lo = lf_limb(C_B, 0), b_hi = lf_limb(C_B, 1);
  orc_lf c_lo = lf_limb(C_C, 0), c_hi = lf_limb(C_C, 1), m_lo = lf_limb(C_M, 0), m_hi = lf_limb(C_M, 1);
  {
    orc_inter* it = &g_cpu[0];
    it->bus = BUS_PROG; it->sign = -1; it->mult = is_real; it->n_el = 10;
    it->el[0] = lf_col(C_PC);
    lf_zero(&it->el[1]);
    for (int k = 1; k <= N_OPS; ++k) lf_add(&it->el[1], C_OP + k - 1, (uint32_t)k);
    it->el[2] = lf_col(C_WR); it->el[3] = lf_col(C_USE2); it->el[4] = lf_col(C_RD); it->el[5] = lf_col(C_RS1);
    it->el[6] = lf_col(C_RS2); it->el[7] = lf_col(C_IMM_LO); it->el[8] = lf_col(C_IMM_HI); it->el[9] = lf_col(C_TGT);
  }
  g_cpu[1] = mem_inter(-1, is_real, lf_col(C_RS1), b_lo, b_hi, lf_col(C_R1_PTS));
  g_cpu[2] = mem_inter(+1, is_real, lf_col(C_RS1), b_lo, b_hi, ts);
  g_cpu[3] = mem_inter(-1, lf_col(C_USE2), lf_col(C_RS2), c_lo, c_hi, lf_col(C_R2_PTS));
  g_cpu[4] = mem_inter(+1, lf_col(C_USE2), lf_col(C_RS2), c_lo, c_hi, lf_plus(ts, 1));
  {
    orc_lf memq, maddr;
    lf_zero(&memq);
    for (int k = OP_LB; k <= OP_SW; ++k) lf_add(&memq, C_OP + k - 1, 1);
    lf_add(&memq, C_OP + OP_ECALL - 1, 1);
    /* word address = (X as a value) - byte offset */
    lf_zero(&maddr);
    for (int i = 0; i < 16; ++i) lf_add(&maddr, C_X + i, 1u << i);
    for (int i = 0; i < 15; ++i) lf_add(&maddr, C_X + 16 + i, (uint32_t)(((uint64_t)65536 << i) % FP));
    /* bit 31 would exceed LF_MAX with the offsets; guest addresses stay below 0x78000000, so it is
     * folded in as 2^31 mod p like the others */
    lf_add(&maddr, C_X + 31, (uint32_t)(((uint64_t)1 << 31) % FP));
    lf_add(&maddr, C_O1, FP - 1); lf_add(&maddr, C_O2, FP - 2); lf_add(&maddr, C_O3, FP - 3);
    g_cpu[5] = mem_inter(-1, memq, maddr, m_lo, m_hi, lf_col(C_M_PTS));
    g_cpu[6] = mem_inter(+1, memq, maddr, lf_col(C_MV_LO), lf_col(C_MV_HI), lf_plus(ts, 2));
    /* the limbs of the four access-time differences are looked up when their access is live */
    for (int j = 0; j < TS_LIMBS; ++j) {
      g_cpu[13 + j] = range_inter(-1, is_real, lf_col(C_R1_D + j));
      g_cpu[15 + j] = range_inter(-1, lf_col(C_USE2), lf_col(C_R2_D + j));
      g_cpu[17 + j] = range_inter(-1, memq, lf_col(C_M_D + j));
      g_cpu[19 + j] = range_inter(-1, lf_col(C_WR), lf_col(C_W_D + j));
    }
  }
  g_cpu[7] = mem_inter(-1, lf_col(C_WR), lf_col(C_RD), lf_col(C_W_PLO), lf_col(C_W_PHI), lf_col(C_W_PTS));
  g_cpu[8] = mem_inter(+1, lf_col(C_WR), lf_col(C_RD), a_lo, a_hi, lf_plus(ts, 3));
  {
    orc_inter* it = &g_cpu[9];
    it->bus = BUS_KCALL; it->sign = +1; it->mult = lf_col(C_OP + OP_KECCAK - 1); it->n_el = 3;
    it->el[0] = ts; it->el[1] = c_lo; it->el[2] = c_hi;
    it = &g_cpu[10];
    it->bus = BUS_MUL; it->sign = +1; it->n_el = 7;
    lf_zero(&it->mult); lf_add(&it->mult, C_OP + OP_MUL - 1, 1); lf_add(&it->mult, C_OP + OP_MULHU - 1, 1);
    it->el[0] = lf_col(C_OP + OP_MULHU - 1);
    it->el[1] = a_lo; it->el[2] = a_hi; it->el[3] = b_lo; it->el[4] = b_hi; it->el[5] = c_lo; it->el[6] = c_hi;
    it = &g_cpu[11];
    it->bus = BUS_PUBC; it->sign = +1; it->n_el = 4;
    lf_zero(&it->mult); lf_add(&it->mult, C_SC + SC_COMMIT, 1); lf_add(&it->mult, C_SC + SC_DEFER, 1);
    lf_zero(&it->el[0]); lf_add(&it->el[0], C_SC + SC_COMMIT, 1); lf_add(&it->el[0], C_SC + SC_DEFER, 2);
    it->el[1] = c_lo; it->el[2] = m_lo; it->el[3] = m_hi;
    it = &g_cpu[12];
    it->bus = BUS_PUBH; it->sign = +1; it->mult = lf_col(C_SC + SC_HALT); it->n_el = 2;
    it->el[0] = c_lo; it->el[1] = c_hi;
  }
  /* keccak chip: on an export row, word i of the input and of the output state, i = 0..49 */
  for (int i = 0; i < 50; ++i) {
    orc_inter* it = &g_keccak[i];
    const int lane = i >> 1, half = i & 1;
    const int out = lane == 0 ? KA_APPP00 + 2 * half : KA_APP + 4 * lane + 2 * half;
    it->bus = BUS_KIO; it->sign = +1; it->mult = lf_col(KA_EXPORT); it->n_el = 6;
    it->el[0] = lf_col(KC_TS); it->el[1] = lf_const((uint32_t)i);
    it->el[2] = lf_col(KA_PREIMAGE + 4 * lane + 2 * half); it->el[3] = lf_col(KA_PREIMAGE + 4 * lane + 2 * half + 1);
    it->el[4] = lf_col(out); it->el[5] = lf_col(out + 1);
  }
  {
    orc_inter* it = &g_kmem[0];
    it->bus = BUS_KCALL; it->sign = -1; it->mult = lf_col(KM_CALL); it->n_el = 3;
    it->el[0] = lf_col(KM_TS); it->el[1] = lf_col(KM_PTR_LO); it->el[2] = lf_col(KM_PTR_HI);
    it = &g_kmem[1];
    it->bus = BUS_KIO; it->sign = -1; it->mult = lf_col(KM_IS_REAL); it->n_el = 6;
    it->el[0] = lf_col(KM_TS); it->el[1] = lf_col(KM_IDX); it->el[2] = lf_col(KM_OLD_LO); it->el[3] = lf_col(KM_OLD_HI);
    it->el[4] = lf_col(KM_NEW_LO); it->el[5] = lf_col(KM_NEW_HI);
    g_kmem[2] = mem_inter(-1, lf_col(KM_IS_REAL), lf_col(KM_ADDR), lf_col(KM_OLD_LO), lf_col(KM_OLD_HI), lf_col(KM_PTS));
    g_kmem[3] = mem_inter(+1, lf_col(KM_IS_REAL), lf_col(KM_ADDR), lf_col(KM_NEW_LO), lf_col(KM_NEW_HI),
                          lf_plus(lf_col(KM_TS), 2));
    for (int j = 0; j < TS_LIMBS; ++j) g_kmem[4 + j] = range_inter(-1, lf_col(KM_IS_REAL), lf_col(KM_D + j));
  }
  g_memfinal[0] = mem_inter(-1, lf_col(MF_IS_REAL), lf_col(MF_ADDR), lf_col(MF_FIN_LO), lf_col(MF_FIN_HI), lf_col(MF_FIN_TS));
  g_memfinal[1] = mem_inter(+1, lf_col(MF_IS_INIT), lf_col(MF_ADDR), lf_limb(MF_INIT, 0), lf_limb(MF_INIT, 1), lf_const(0));
  /* image / program: the row is [preprocessed | main] */
  g_image[0] = mem_inter(+1, lf_col(IMAGE_PREP_WIDTH + 0), lf_col(IMG_P_ADDR), lf_col(IMG_P_LO), lf_col(IMG_P_HI), lf_const(0));
  {
    orc_inter* it = &g_program[0];
    it->bus = BUS_PROG; it->sign = +1; it->mult = lf_col(PROGRAM_PREP_WIDTH + 0); it->n_el = 10;
    for (int j = 0; j < 10; ++j) it->el[j] = lf_col(j);
  }
  for (int hi = 0; hi < 2; ++hi) {
    orc_inter* it = &g_mul[hi];
    it->bus = BUS_MUL; it->sign = -1; it->n_el = 7;
    if (hi) it->mult = lf_col(MU_HI);
    else { lf_zero(&it->mult); lf_add(&it->mult, MU_IS_REAL, 1); lf_add(&it->mult, MU_HI, FP - 1); }
    it->el[0] = lf_const((uint32_t)hi);
    it->el[1] = lf_limb(MU_P, 2 * hi); it->el[2] = lf_limb(MU_P, 2 * hi + 1);
    it->el[3] = lf_limb(MU_B, 0); it->el[4] = lf_limb(MU_B, 1); it->el[5] = lf_limb(MU_C, 0); it->el[6] = lf_limb(MU_C, 1);
  }
  g_range[0] = range_inter(+1, lf_col(RANGE_PREP_WIDTH + 0), lf_col(0));
  g_chips[CH_RANGE] = (orc_chip){"range", RANGE_PREP_WIDTH, RANGE_WIDTH, 1, g_range, 0};
  g_chips[CH_CPU] = (orc_chip){"cpu", 0, CPU_WIDTH, 21, g_cpu, 0};
  g_chips[CH_CPU2] = (orc_chip){"cpu2", 0, CPU_WIDTH, 21, g_cpu, 0};
  g_chips[CH_KECCAK] = (orc_chip){"keccak", 0, KECCAK_WIDTH, 50, g_keccak, 0};
  g_chips[CH_KMEM] = (orc_chip){"keccak-mem", 0, KMEM_WIDTH, 6, g_kmem, 0};
  g_chips[CH_MEMFINAL] = (orc_chip){"mem-final", 0, MEMFINAL_WIDTH, 2, g_memfinal, 0};
  g_chips[CH_IMAGE] = (orc_chip){"image", IMAGE_PREP_WIDTH, IMAGE_WIDTH, 1, g_image, 0};
  g_chips[CH_PROGRAM] = (orc_chip){"program", PROGRAM_PREP_WIDTH, PROGRAM_WIDTH, 1, g_program, 0};
  g_chips[CH_MUL] = (orc_chip){"mul", 0, MUL_WIDTH, 2, g_mul, 0};
  g_ready = 1;
  for (int c = 0; c < N_CHIPS; ++c) g_chips[c].n_constraints = count_constraints(c);
}

const orc_chip* orc_machine_chip(int chip) {
  build();
  return &g_chips[chip];
}

/* ------------------------------------------------------------------------------------------
 * heights and trace generation
 * ---------------------------------------------------------------------------------------- */
static int clog2(size_t v) {
  int l = 0;
  while (((size_t)1 << l) < v) ++l;
  return l;
}
static int at_least5(int l) { return l < 5 ? 5 : l; }

/* rows of the first CPU instance: the largest power of two strictly below the cycle count (at least 32) */
static size_t cpu_split(size_t n_cycles) {
  size_t h0 = 32;
  while (2 * h0 < n_cycles) h0 *= 2;
  return h0;
}

void orc_machine_cpu_pub(const orc_machine_input* in, int chip, uint32_t pub[CPUPUB_N]) {
  const size_t h0 = cpu_split(in->n_cycles);
  const uint32_t handover = h0 < in->n_cycles ? in->cycles[12 * h0] : 0;
  if (chip == CH_CPU) {
    pub[CPUPUB_START_PC] = in->entry; pub[CPUPUB_START_TS] = 4; pub[CPUPUB_HAS_SUCC] = 1; pub[CPUPUB_END_PC] = handover;
  } else {
    pub[CPUPUB_START_PC] = handover; pub[CPUPUB_START_TS] = 4 * ((uint32_t)h0 + 1); pub[CPUPUB_HAS_SUCC] = 0; pub[CPUPUB_END_PC] = 0;
  }
}

void orc_machine_heights(const orc_machine_input* in, int logh[N_CHIPS]) {
  const size_t h0 = cpu_split(in->n_cycles);
  logh[CH_CPU] = clog2(h0);
  logh[CH_CPU2] = at_least5(clog2(in->n_cycles > h0 ? in->n_cycles - h0 : 1));
  logh[CH_KECCAK] = at_least5(clog2(24 * in->n_keccak));
  logh[CH_KMEM] = at_least5(clog2(50 * in->n_keccak));
  logh[CH_MEMFINAL] = at_least5(clog2(in->n_memfinal));
  logh[CH_IMAGE] = in->log_image;
  logh[CH_PROGRAM] = in->log_prog;
  logh[CH_MUL] = at_least5(clog2(in->n_muls));
  logh[CH_RANGE] = RANGE_LOG_H;
}

static void put_bits(uint32_t* t, size_t h, size_t r, int col, uint32_t v, int n) {
  for (int i = 0; i < n; ++i) t[(size_t)(col + i) * h + r] = (v >> i) & 1u;
}

typedef struct { uint32_t ts, ptr; uint64_t in[25]; uint32_t pts[50]; } kcall_t;

/* an access-time difference as its two limbs */
static void put_gap(uint32_t* t, size_t h, size_t r, int col, uint32_t gap) {
  t[(size_t)col * h + r] = gap & ((1u << TS_LIMB_BITS) - 1);
  t[(size_t)(col + 1) * h + r] = gap >> TS_LIMB_BITS;
}
/* the four differences of one cycle (0 where the access is not live), as fill_cpu writes them */
static void cycle_gaps(const orc_machine_input* in, size_t r, uint32_t gap[4], int live[4]) {
  const uint32_t* cy = in->cycles + 12 * r;
  const uint32_t* p = in->program + 9 * (size_t)((cy[0] - in->text_base) >> 2);
  const uint32_t op = p[1], ts = 4 * ((uint32_t)r + 1);
  live[0] = 1; gap[0] = ts - cy[7] - 1;
  live[1] = p[3] != 0; gap[1] = ts - cy[8];
  live[2] = (op >= OP_LB && op <= OP_SW) || op == OP_ECALL; gap[2] = ts + 1 - cy[9];
  live[3] = p[2] != 0; gap[3] = ts + 2 - cy[10];
}

/* rows [0, h) of the instance whose first row is cycle `row0` */
static void fill_cpu(const orc_machine_input* in, size_t h, uint32_t* t, size_t row0) {
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < h; ++r) {
#define T(col) t[(size_t)(col) * h + r]
    const size_t g = row0 + r;  /* cycle index */
    const uint32_t ts = 4 * ((uint32_t)g + 1);
    T(C_TS) = ts;
    if (g >= in->n_cycles) continue;
    const uint32_t* cy = in->cycles + 12 * g;
    const uint32_t pc = cy[0], a = cy[1], b = cy[2], c = cy[3], m = cy[4], mv = cy[5], wprev = cy[6];
    const uint32_t* p = in->program + 9 * (size_t)((pc - in->text_base) >> 2);
    const uint32_t op = p[1], wr = p[2], use2 = p[3], rd = p[4], rs1 = p[5], rs2 = p[6], imm = p[7], tgt = p[8];
    T(C_IS_REAL) = 1; T(C_PC) = pc;
    T(C_OP + op - 1) = 1;
    T(C_WR) = wr; T(C_USE2) = use2; T(C_RD) = rd; T(C_RS1) = rs1; T(C_RS2) = rs2;
    T(C_IMM_LO) = imm & 0xffff; T(C_IMM_HI) = imm >> 16; T(C_TGT) = tgt;
    T(C_A) = a & 0xffff; T(C_A + 1) = a >> 16; put_bits(t, h, r, C_B, b, 32); put_bits(t, h, r, C_C, c, 32); put_bits(t, h, r, C_M, m, 32);
    T(C_MV_LO) = mv & 0xffff; T(C_MV_HI) = mv >> 16;
    uint32_t x = 0, next = pc + 4;
    const uint32_t blo = b & 0xffff, bhi = b >> 16, clo = c & 0xffff, chi = c >> 16, alo = a & 0xffff, ahi = a >> 16;
    switch (op) {
      case OP_ADD: {
        uint32_t k0 = (blo + clo) >> 16;
        const uint32_t k1 = (bhi + chi + k0) >> 16;
        T(C_K0) = k0; T(C_K1) = k1;
        /* soundness tests: ZKSP_ORACLE_NONCANON=<row> makes that addition claim the other carry, i.e. write the
         * same sum with limbs out of range.  The row's constraints still hold; nothing can read the value back. */
        const char* nc = getenv("ZKSP_ORACLE_NONCANON");
        if (nc && r == (size_t)strtoull(nc, NULL, 10)) {
          k0 ^= 1u;
          T(C_K0) = k0;
          T(C_A) = f_sub(f_add(blo, clo), k0 ? 65536u : 0u);
          T(C_A + 1) = f_sub(f_add(f_add(bhi, chi), k0), k1 ? 65536u : 0u);
        }
        break;
      }
      case OP_SUB: { uint32_t k0 = (alo + clo) >> 16; T(C_K0) = k0; T(C_K1) = (ahi + chi + k0) >> 16; break; }
      case OP_SLL: case OP_SRL: case OP_SRA: x = 1u << (c & 31); break;
      case OP_SLT: case OP_SLTU: case OP_BEQ: case OP_BNE: case OP_BLT: case OP_BGE: case OP_BLTU: case OP_BGEU: {
        const int sgn = (op == OP_SLT || op == OP_BLT || op == OP_BGE);
        const uint32_t k0 = blo < clo;
        const uint32_t dlo = blo - clo + 65536 * k0;
        const uint32_t lt = sgn ? ((int32_t)b < (int32_t)c) : (b < c);
        const int64_t dhi = (int64_t)bhi - chi - k0 + 65536 * (int64_t)lt + (sgn ? 65536 * ((int64_t)(c >> 31) - (int64_t)(b >> 31)) : 0);
        x = dlo | ((uint32_t)dhi << 16);
        T(C_K0) = k0; T(C_K1) = lt;
        if (op == OP_BEQ || op == OP_BNE) {
          const uint32_t z = dlo + (uint32_t)dhi;
          T(C_EQ) = z == 0; T(C_INV) = z ? f_inv(z) : 0;
          if ((op == OP_BEQ) == (z == 0)) next = tgt;
        } else if (op == OP_BLT || op == OP_BLTU) { if (lt) next = tgt; }
        else if (op == OP_BGE || op == OP_BGEU) { if (!lt) next = tgt; }
        break;
      }
      case OP_JAL: next = tgt; break;
      case OP_JALR: case OP_LB: case OP_LH: case OP_LW: case OP_LBU: case OP_LHU: case OP_SB: case OP_SH: case OP_SW: {
        const uint32_t ilo = imm & 0xffff, ihi = imm >> 16;
        const uint32_t k2 = (blo + ilo) >> 16;
        T(C_K2) = k2; T(C_K3) = (bhi + ihi + k2) >> 16;
        x = b + imm;
        if (op == OP_JALR) next = x & ~1u;
        else T(C_O0 + (x & 3)) = 1;
        break;
      }
      case OP_ECALL: {
        x = 11; T(C_O0) = 1;
        static const uint32_t codes[6] = {0x00, 0x02, 0x10, 0x1a, 0xf0, 0xf1};
        for (int k = 0; k < 6; ++k) if (b == codes[k]) T(C_SC + k) = 1;
        break;
      }
      case OP_KECCAK: next = b; break;
      default: break;
    }
    put_bits(t, h, r, C_X, x, 32);
    T(C_NEXT_PC) = next;
    uint32_t gap[4];
    int live[4];
    cycle_gaps(in, g, gap, live);
    T(C_R1_PTS) = cy[7]; put_gap(t, h, r, C_R1_D, gap[0]);
    if (live[1]) { T(C_R2_PTS) = cy[8]; put_gap(t, h, r, C_R2_D, gap[1]); }
    if (live[2]) { T(C_M_PTS) = cy[9]; put_gap(t, h, r, C_M_D, gap[2]); }
    if (live[3]) {
      T(C_W_PTS) = cy[10]; put_gap(t, h, r, C_W_D, gap[3]);
      T(C_W_PLO) = wprev & 0xffff; T(C_W_PHI) = wprev >> 16;
    }
#undef T
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
    case CH_CPU: fill_cpu(in, h, t, 0); break;
    case CH_CPU2: fill_cpu(in, h, t, cpu_split(in->n_cycles)); break;
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
        put_gap(t, h, r, KM_D, k->ts + 1 - k->pts[i]);
      }
      break;
    case CH_MEMFINAL:
      for (size_t r = 0; r < in->n_memfinal; ++r) {
        const uint32_t* f = in->memfinal + 5 * r;
        T(MF_IS_REAL) = 1; T(MF_ADDR) = f[0]; T(MF_IS_INIT) = f[4];
        T(MF_FIN_LO) = f[2] & 0xffff; T(MF_FIN_HI) = f[2] >> 16; T(MF_FIN_TS) = f[3];
        if (r + 1 < in->n_memfinal) put_bits(t, h, r, MF_DIFF, in->memfinal[5 * (r + 1)] - f[0] - 1, 32);
        if (f[4]) put_bits(t, h, r, MF_INIT, f[1], 32);
      }
      break;
    case CH_IMAGE:
      for (size_t r = 0; r < in->n_image; ++r) {
        prep[(size_t)IMG_P_ADDR * h + r] = in->image[2 * r];
        prep[(size_t)IMG_P_LO * h + r] = in->image[2 * r + 1] & 0xffff;
        prep[(size_t)IMG_P_HI * h + r] = in->image[2 * r + 1] >> 16;
        if (in->image_used) T(0) = in->image_used[r];
      }
      break;
    case CH_PROGRAM:
      for (size_t r = 0; r < in->n_program; ++r) {
        const uint32_t* p = in->program + 9 * r;
        uint32_t* q = prep + r;
        q[(size_t)PR_PC * h] = p[0]; q[(size_t)PR_OP * h] = p[1]; q[(size_t)PR_WR * h] = p[2]; q[(size_t)PR_USE2 * h] = p[3];
        q[(size_t)PR_RD * h] = p[4]; q[(size_t)PR_RS1 * h] = p[5]; q[(size_t)PR_RS2 * h] = p[6];
        q[(size_t)PR_IMM_LO * h] = p[7] & 0xffff; q[(size_t)PR_IMM_HI * h] = p[7] >> 16; q[(size_t)PR_TGT * h] = p[8];
        if (in->prog_mult) T(0) = in->prog_mult[r] % FP;
      }
      break;
    case CH_MUL:
      for (size_t r = 0; r < in->n_muls; ++r) {
        const uint32_t* mu = in->muls + 3 * r;
        const uint32_t b = mu[1], c = mu[2];
        const uint64_t prod = (uint64_t)b * c;
        T(MU_IS_REAL) = 1; T(MU_HI) = mu[0];
        put_bits(t, h, r, MU_B, b, 32); put_bits(t, h, r, MU_C, c, 32);
        put_bits(t, h, r, MU_P, (uint32_t)prod, 32); put_bits(t, h, r, MU_P + 32, (uint32_t)(prod >> 32), 32);
        uint64_t s[7] = {0};
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j) s[i + j] += (uint64_t)((b >> (8 * i)) & 0xff) * ((c >> (8 * j)) & 0xff);
        const uint64_t q0 = (s[0] + 256 * s[1]) >> 16, q1 = (s[2] + 256 * s[3] + q0) >> 16, q2 = (s[4] + 256 * s[5] + q1) >> 16;
        put_bits(t, h, r, MU_Q0, (uint32_t)q0, 10); put_bits(t, h, r, MU_Q1, (uint32_t)q1, 11); put_bits(t, h, r, MU_Q2, (uint32_t)q2, 10);
      }
      break;
    case CH_RANGE: {
      const uint32_t mask = (1u << TS_LIMB_BITS) - 1;
      for (size_t r = 0; r < h; ++r) prep[r] = (uint32_t)r;
      /* multiplicity of a value = the live limbs equal to it, over the CPU and keccak-memory chips */
      for (size_t r = 0; r < in->n_cycles; ++r) {
        uint32_t gap[4];
        int live[4];
        cycle_gaps(in, r, gap, live);
        for (int q = 0; q < 4; ++q)
          if (live[q] && (gap[q] >> TS_LIMB_BITS) <= mask) { t[gap[q] & mask]++; t[gap[q] >> TS_LIMB_BITS]++; }
      }
      for (size_t p = 0; p < in->n_keccak; ++p) {
        const kcall_t* k = (const kcall_t*)(in->keccak + 408 * p);
        for (int i = 0; i < 50; ++i) {
          const uint32_t gap = k->ts + 1 - k->pts[i];
          if ((gap >> TS_LIMB_BITS) <= mask) { t[gap & mask]++; t[gap >> TS_LIMB_BITS]++; }
        }
      }
      break;
    }
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
/* an access-time difference from its two range-checked limbs */
static inline fe gap_val(const uint32_t* row, int col) { return f_add(row[col], f_mul(1u << TS_LIMB_BITS, row[col + 1])); }
#define F65536 65536u

static void cpu_constraints(const uint32_t* l, const uint32_t* n, fe is_first, fe is_last, fe is_trans, const uint32_t* pub,
                            sink* s) {
  const fe one = 1;
  /* ---- booleans ---- */
  emit(s, bool_c(l[C_IS_REAL]));
  for (int k = 0; k < N_OPS; ++k) emit(s, bool_c(l[C_OP + k]));
  emit(s, bool_c(l[C_WR]));
  emit(s, bool_c(l[C_USE2]));
  for (int i = 0; i < 128; ++i) emit(s, bool_c(l[C_B + i])); /* B, C, M, X */
  for (int i = 0; i < 4; ++i) emit(s, bool_c(l[C_K0 + i]));
  emit(s, bool_c(l[C_EQ]));
  for (int i = 0; i < 4; ++i) emit(s, bool_c(l[C_O0 + i]));
  for (int i = 0; i < 6; ++i) emit(s, bool_c(l[C_SC + i]));
  /* ---- row structure ---- */
#define OPF(k) l[C_OP + (k) - 1]
  const fe is_real = l[C_IS_REAL];
  fe opsum = 0;
  for (int k = 0; k < N_OPS; ++k) opsum = f_add(opsum, l[C_OP + k]);
  emit(s, f_sub(opsum, is_real));
  emit(s, f_mul(l[C_WR], f_sub(one, is_real)));
  emit(s, f_mul(l[C_USE2], f_sub(one, is_real)));
  emit(s, f_mul(is_first, f_sub(is_real, one)));
  emit(s, f_mul(is_first, f_sub(l[C_PC], pub[CPUPUB_START_PC] % FP)));
  emit(s, f_mul(is_first, f_sub(l[C_TS], pub[CPUPUB_START_TS] % FP)));
  emit(s, f_mul(is_trans, f_sub(f_sub(n[C_TS], l[C_TS]), 4)));
  emit(s, f_mul(f_mul(is_trans, n[C_IS_REAL]), f_sub(n[C_PC], l[C_NEXT_PC])));
  emit(s, f_mul(is_trans, f_add(f_sub(n[C_IS_REAL], is_real), l[C_SC + SC_HALT])));
  fe scsum = 0;
  for (int k = 0; k < 6; ++k) scsum = f_add(scsum, l[C_SC + k]);
  emit(s, f_sub(scsum, OPF(OP_ECALL)));
  /* ---- limbs ---- */
  const fe a_lo = l[C_A], a_hi = l[C_A + 1], b_lo = limb_of(l, C_B, 0), b_hi = limb_of(l, C_B, 1);
  const fe c_lo = limb_of(l, C_C, 0), c_hi = limb_of(l, C_C, 1), m_lo = limb_of(l, C_M, 0), m_hi = limb_of(l, C_M, 1);
  const fe x_lo = limb_of(l, C_X, 0), x_hi = limb_of(l, C_X, 1);
  const fe k0 = l[C_K0], k1 = l[C_K1], k2 = l[C_K2], k3 = l[C_K3];
  /* ---- operand C is the immediate ---- */
  const fe immc = f_sub(is_real, l[C_USE2]);
  emit(s, f_mul(immc, f_sub(c_lo, l[C_IMM_LO])));
  emit(s, f_mul(immc, f_sub(c_hi, l[C_IMM_HI])));
  /* ---- add / sub ---- */
  emit(s, f_mul(OPF(OP_ADD), f_sub(f_add(b_lo, c_lo), f_add(a_lo, f_mul(F65536, k0)))));
  emit(s, f_mul(OPF(OP_ADD), f_sub(f_add(f_add(b_hi, c_hi), k0), f_add(a_hi, f_mul(F65536, k1)))));
  emit(s, f_mul(OPF(OP_SUB), f_sub(f_add(a_lo, c_lo), f_add(b_lo, f_mul(F65536, k0)))));
  emit(s, f_mul(OPF(OP_SUB), f_sub(f_add(f_add(a_hi, c_hi), k0), f_add(b_hi, f_mul(F65536, k1)))));
  /* ---- bitwise ---- */
  for (int op = OP_XOR; op <= OP_AND; ++op)
    for (int h = 0; h < 2; ++h) {
      fe acc = 0;
      for (int i = 15; i >= 0; --i) {
        const fe b = l[C_B + 16 * h + i], c = l[C_C + 16 * h + i], bc = f_mul(b, c);
        fe bit = op == OP_AND ? bc : op == OP_OR ? f_sub(f_add(b, c), bc) : f_sub(f_add(b, c), f_add(bc, bc));
        acc = f_add(f_add(acc, acc), bit);
      }
      emit(s, f_mul(OPF(op), f_sub(h ? a_hi : a_lo, acc)));
    }
  /* ---- shifts: X is the one-hot of the amount ---- */
  {
    const fe sh = f_add(f_add(OPF(OP_SLL), OPF(OP_SRL)), OPF(OP_SRA));
    fe sum = 0, idx = 0;
    for (int k = 0; k < 32; ++k) { sum = f_add(sum, l[C_X + k]); idx = f_add(idx, f_mul((fe)k, l[C_X + k])); }
    emit(s, f_mul(sh, f_sub(sum, one)));
    emit(s, f_mul(sh, f_sub(idx, bits_val(l, C_C, 5))));
    for (int kind = 0; kind < 3; ++kind) {
      fe t[32];
      for (int j = 0; j < 32; ++j) {
        fe acc = 0;
        for (int k = 0; k < 32; ++k) {
          int src;
          if (kind == 0) { if (k > j) continue; src = j - k; }
          else if (kind == 1) { if (j + k > 31) continue; src = j + k; }
          else src = j + k > 31 ? 31 : j + k;
          acc = f_add(acc, f_mul(l[C_X + k], l[C_B + src]));
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
  /* ---- comparisons: X = B - C (mod 2^32, the sign bits flipped for signed orders), K1 = "less than" ---- */
  {
    fe cmp = 0, sgn = f_add(f_add(OPF(OP_SLT), OPF(OP_BLT)), OPF(OP_BGE));
    const int ops[8] = {OP_SLT, OP_SLTU, OP_BEQ, OP_BNE, OP_BLT, OP_BGE, OP_BLTU, OP_BGEU};
    for (int i = 0; i < 8; ++i) cmp = f_add(cmp, OPF(ops[i]));
    emit(s, f_mul(cmp, f_sub(f_add(f_sub(b_lo, c_lo), f_mul(F65536, k0)), x_lo)));
    emit(s, f_add(f_mul(cmp, f_sub(f_add(f_sub(f_sub(b_hi, c_hi), k0), f_mul(F65536, k1)), x_hi)),
                  f_mul(F65536, f_mul(sgn, f_sub(l[C_C + 31], l[C_B + 31])))));
    const fe bq = f_add(OPF(OP_BEQ), OPF(OP_BNE)), z = f_add(x_lo, x_hi);
    emit(s, f_mul(bq, f_add(f_sub(f_mul(z, l[C_INV]), one), l[C_EQ])));
    emit(s, f_mul(bq, f_mul(z, l[C_EQ])));
    const fe slt = f_add(OPF(OP_SLT), OPF(OP_SLTU));
    emit(s, f_mul(slt, f_sub(a_lo, k1)));
    emit(s, f_mul(slt, a_hi));
  }
  /* ---- next pc ---- */
  {
    const fe pc4 = f_add(l[C_PC], 4), np = l[C_NEXT_PC], tgt = l[C_TGT];
    fe def = is_real;
    const int nd[9] = {OP_JAL, OP_JALR, OP_BEQ, OP_BNE, OP_BLT, OP_BGE, OP_BLTU, OP_BGEU, OP_KECCAK};
    for (int i = 0; i < 9; ++i) def = f_sub(def, OPF(nd[i]));
    emit(s, f_mul(def, f_sub(np, pc4)));
    emit(s, f_mul(OPF(OP_JAL), f_sub(np, tgt)));
    emit(s, f_mul(OPF(OP_JAL), f_sub(a_lo, c_lo)));
    emit(s, f_mul(OPF(OP_JAL), f_sub(a_hi, c_hi)));
    emit(s, f_mul(OPF(OP_JALR), f_sub(f_add(a_lo, f_mul(F65536, a_hi)), tgt)));
    emit(s, f_mul(OPF(OP_JALR), f_sub(np, f_sub(f_add(x_lo, f_mul(F65536, x_hi)), l[C_X]))));
    const fe eq = l[C_EQ];
    /* taken -> tgt, else pc + 4:  np - pc4 - taken * (tgt - pc4) */
    const fe d = f_sub(tgt, pc4), base = f_sub(np, pc4);
    emit(s, f_mul(OPF(OP_BEQ), f_sub(base, f_mul(eq, d))));
    emit(s, f_mul(OPF(OP_BNE), f_sub(base, f_mul(f_sub(one, eq), d))));
    emit(s, f_mul(OPF(OP_BLT), f_sub(base, f_mul(k1, d))));
    emit(s, f_mul(OPF(OP_BGE), f_sub(base, f_mul(f_sub(one, k1), d))));
    emit(s, f_mul(OPF(OP_BLTU), f_sub(base, f_mul(k1, d))));
    emit(s, f_mul(OPF(OP_BGEU), f_sub(base, f_mul(f_sub(one, k1), d))));
    emit(s, f_mul(OPF(OP_KECCAK), f_sub(np, f_add(b_lo, f_mul(F65536, b_hi)))));
  }
  /* ---- address adder: X = B + imm ---- */
  fe loads = 0, stores = 0;
  for (int k = OP_LB; k <= OP_LHU; ++k) loads = f_add(loads, OPF(k));
  for (int k = OP_SB; k <= OP_SW; ++k) stores = f_add(stores, OPF(k));
  {
    const fe ad = f_add(f_add(loads, stores), OPF(OP_JALR));
    emit(s, f_mul(ad, f_sub(f_add(b_lo, l[C_IMM_LO]), f_add(x_lo, f_mul(F65536, k2)))));
    emit(s, f_mul(ad, f_sub(f_add(f_add(b_hi, l[C_IMM_HI]), k2), f_add(x_hi, f_mul(F65536, k3)))));
  }
  /* ---- byte offset one-hot ---- */
  const fe o0 = l[C_O0], o1 = l[C_O1], o2 = l[C_O2], o3 = l[C_O3];
  {
    const fe ls = f_add(loads, stores);
    emit(s, f_mul(ls, f_sub(f_add(f_add(o0, o1), f_add(o2, o3)), one)));
    emit(s, f_mul(ls, f_sub(f_add(f_add(o1, f_add(o2, o2)), f_mul(3, o3)), f_add(l[C_X], f_add(l[C_X + 1], l[C_X + 1])))));
    emit(s, f_mul(OPF(OP_ECALL), f_sub(o0, one)));
    emit(s, f_mul(OPF(OP_ECALL), f_add(f_add(o1, o2), o3)));
    emit(s, f_mul(OPF(OP_ECALL), f_sub(x_lo, 11)));
    emit(s, f_mul(OPF(OP_ECALL), x_hi));
  }
  /* ---- loads ---- */
  {
    const fe mb[4] = {byte_of(l, C_M, 0), byte_of(l, C_M, 1), byte_of(l, C_M, 2), byte_of(l, C_M, 3)};
    emit(s, f_mul(OPF(OP_LW), f_sub(o0, one)));
    emit(s, f_mul(OPF(OP_LW), f_sub(a_lo, m_lo)));
    emit(s, f_mul(OPF(OP_LW), f_sub(a_hi, m_hi)));
    const fe hv = f_add(f_mul(o0, m_lo), f_mul(o2, m_hi)), hs = f_add(f_mul(o0, l[C_M + 15]), f_mul(o2, l[C_M + 31]));
    emit(s, f_mul(OPF(OP_LHU), f_add(o1, o3)));
    emit(s, f_mul(OPF(OP_LHU), f_sub(a_lo, hv)));
    emit(s, f_mul(OPF(OP_LHU), a_hi));
    emit(s, f_mul(OPF(OP_LH), f_add(o1, o3)));
    emit(s, f_mul(OPF(OP_LH), f_sub(a_lo, hv)));
    emit(s, f_mul(OPF(OP_LH), f_sub(a_hi, f_mul(65535, hs))));
    fe bv = 0, bs = 0;
    for (int p = 0; p < 4; ++p) { bv = f_add(bv, f_mul(l[C_O0 + p], mb[p])); bs = f_add(bs, f_mul(l[C_O0 + p], l[C_M + 8 * p + 7])); }
    emit(s, f_mul(OPF(OP_LBU), f_sub(a_lo, bv)));
    emit(s, f_mul(OPF(OP_LBU), a_hi));
    emit(s, f_mul(OPF(OP_LB), f_sub(a_lo, f_add(bv, f_mul(0xff00, bs)))));
    emit(s, f_mul(OPF(OP_LB), f_sub(a_hi, f_mul(65535, bs))));
    const fe keep = f_add(loads, OPF(OP_ECALL));
    emit(s, f_mul(keep, f_sub(l[C_MV_LO], m_lo)));
    emit(s, f_mul(keep, f_sub(l[C_MV_HI], m_hi)));
    /* ---- stores ---- */
    emit(s, f_mul(OPF(OP_SW), f_sub(o0, one)));
    emit(s, f_mul(OPF(OP_SW), f_sub(l[C_MV_LO], c_lo)));
    emit(s, f_mul(OPF(OP_SW), f_sub(l[C_MV_HI], c_hi)));
    emit(s, f_mul(OPF(OP_SH), f_add(o1, o3)));
    emit(s, f_mul(OPF(OP_SH), f_sub(f_sub(l[C_MV_LO], m_lo), f_mul(o0, f_sub(c_lo, m_lo)))));
    emit(s, f_mul(OPF(OP_SH), f_sub(f_sub(l[C_MV_HI], m_hi), f_mul(o2, f_sub(c_lo, m_hi)))));
    const fe cb = byte_of(l, C_C, 0);
    emit(s, f_mul(OPF(OP_SB), f_sub(f_sub(l[C_MV_LO], m_lo),
                                    f_add(f_mul(o0, f_sub(cb, mb[0])), f_mul(256, f_mul(o1, f_sub(cb, mb[1])))))));
    emit(s, f_mul(OPF(OP_SB), f_sub(f_sub(l[C_MV_HI], m_hi),
                                    f_add(f_mul(o2, f_sub(cb, mb[2])), f_mul(256, f_mul(o3, f_sub(cb, mb[3])))))));
  }
  /* ---- ecall ---- */
  {
    static const uint32_t codes[6] = {0x00, 0x02, 0x10, 0x1a, 0xf0, 0xf1};
    fe code = 0;
    for (int k = 0; k < 6; ++k) code = f_add(code, f_mul(codes[k], l[C_SC + k]));
    emit(s, f_mul(OPF(OP_ECALL), f_sub(b_lo, code)));
    emit(s, f_mul(OPF(OP_ECALL), b_hi));
    const fe same = f_sub(OPF(OP_ECALL), l[C_SC + SC_HINT_LEN]);
    emit(s, f_mul(same, f_sub(a_lo, b_lo)));
    emit(s, f_mul(same, f_sub(a_hi, b_hi)));
  }
  /* ---- previous access times are older ---- */
  {
    fe memq = f_add(f_add(loads, stores), OPF(OP_ECALL));
    const fe ts = l[C_TS];
    emit(s, f_mul(is_real, f_sub(f_sub(f_sub(ts, l[C_R1_PTS]), one), gap_val(l, C_R1_D))));
    emit(s, f_mul(l[C_USE2], f_sub(f_sub(ts, l[C_R2_PTS]), gap_val(l, C_R2_D))));
    emit(s, f_mul(memq, f_sub(f_sub(f_add(ts, one), l[C_M_PTS]), gap_val(l, C_M_D))));
    emit(s, f_mul(l[C_WR], f_sub(f_sub(f_add(ts, 2), l[C_W_PTS]), gap_val(l, C_W_D))));
  }
  /* ---- hand-over to the next instance: its last row is a real row that does not halt, and names the pc the next
   * instance starts at ---- */
  {
    const fe succ = f_mul(is_last, pub[CPUPUB_HAS_SUCC] % FP);
    emit(s, f_mul(succ, f_add(f_sub(one, is_real), l[C_SC + SC_HALT])));
    emit(s, f_mul(succ, f_sub(l[C_NEXT_PC], pub[CPUPUB_END_PC] % FP)));
  }
#undef OPF
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
  emit(s, f_mul(l[KM_IS_REAL], f_sub(f_sub(f_add(l[KM_TS], one), l[KM_PTS]), gap_val(l, KM_D))));
}

static void memfinal_constraints(const uint32_t* l, const uint32_t* n, fe is_trans, sink* s) {
  const fe one = 1;
  emit(s, bool_c(l[MF_IS_REAL])); emit(s, bool_c(l[MF_IS_INIT]));
  for (int i = 0; i < 64; ++i) emit(s, bool_c(l[MF_DIFF + i])); /* DIFF, INIT */
  emit(s, f_mul(l[MF_IS_INIT], f_sub(one, l[MF_IS_REAL])));
  const fe tn = f_mul(is_trans, n[MF_IS_REAL]);
  emit(s, f_mul(tn, f_sub(one, l[MF_IS_REAL])));
  /* 32-bit difference as a field element: addresses stay below 0x78000000 < p */
  fe diff = 0;
  for (int i = 31; i >= 0; --i) diff = f_add(f_add(diff, diff), l[MF_DIFF + i]);
  emit(s, f_mul(tn, f_sub(f_sub(f_sub(n[MF_ADDR], l[MF_ADDR]), one), diff)));
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
}

static void run_constraints(int chip, const uint32_t* prep, const uint32_t* loc, const uint32_t* nxt, uint32_t is_first,
                            uint32_t is_last, uint32_t is_trans, const uint32_t* pub, sink* s) {
  (void)prep;
  static const uint32_t no_pub[CPUPUB_N] = {0, 0, 0, 0};
  switch (chip) {
    case CH_CPU:
    case CH_CPU2: cpu_constraints(loc, nxt, is_first, is_last, is_trans, pub ? pub : no_pub, s); break;
    case CH_KECCAK:
      if (s->out) orc_keccak_constraints(loc, nxt, is_first, is_last, is_trans, s->out + s->k);
      s->k += KA_NUM_CONSTRAINTS;
      /* the call time is constant inside a permutation's 24 rows */
      emit(s, f_mul(f_mul(is_trans, f_sub(1, loc[KA_FLAGS + 23])), f_sub(nxt[KC_TS], loc[KC_TS])));
      break;
    case CH_KMEM: kmem_constraints(loc, nxt, is_first, is_trans, s); break;
    case CH_MEMFINAL: memfinal_constraints(loc, nxt, is_trans, s); break;
    case CH_IMAGE: emit(s, bool_c(loc[0])); break;
    case CH_PROGRAM: break;
    case CH_MUL: mul_constraints(loc, s); break;
    case CH_RANGE: break;
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
