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
/* The execution is split over two instances of the CPU chip: cycles [0, H0) in CH_CPU with H0 the largest power of two
 * below the cycle count, the rest in CH_CPU2 (a power of two again): 391 400 cycles take 2^18 + 2^17 rows, not 2^19. */
enum { CH_CPU = 0, CH_KECCAK, CH_KMEM, CH_MEMFINAL, CH_IMAGE, CH_PROGRAM, CH_MUL, CH_RANGE, CH_CPU2, N_CHIPS };

/* ---- AIR opcodes (Program table column OP; CPU selector k-1) ---- */
enum {
  OP_ADD = 1, OP_SUB, OP_XOR, OP_OR, OP_AND, OP_SLL, OP_SRL, OP_SRA, OP_SLT, OP_SLTU, OP_JAL, OP_JALR, OP_BEQ, OP_BNE,
  OP_BLT, OP_BGE, OP_BLTU, OP_BGEU, OP_LB, OP_LH, OP_LW, OP_LBU, OP_LHU, OP_SB, OP_SH, OP_SW, OP_MUL, OP_MULHU,
  OP_ECALL, OP_KECCAK, N_OPS_P1
};
#define N_OPS 30
/* access-time differences: two limbs of TS_LIMB_BITS bits, each looked up in the range table */
#define TS_LIMB_BITS 12
#define TS_LIMBS 2

/* ---- CPU chip main columns ---- */
enum {
  C_IS_REAL = 0, C_PC, C_TS, C_NEXT_PC,
  C_OP = 4,                    /* 30 selectors: column C_OP + (op - 1) */
  C_WR = C_OP + N_OPS, C_USE2, C_RD, C_RS1, C_RS2, C_IMM_LO, C_IMM_HI, C_TGT,
  C_A, /* value written: two 16-bit limbs (read back only through the bits of B, C or M) */
  C_B = C_A + 2, C_C = C_B + 32, C_M = C_C + 32, C_X = C_M + 32,
  C_MV_LO = C_X + 32, C_MV_HI,
  C_K0, C_K1, C_K2, C_K3, C_EQ, C_INV,
  C_O0, C_O1, C_O2, C_O3,
  C_SC,                        /* 6 syscall flags: HALT, WRITE, COMMIT, DEFER, HINT_LEN, HINT_READ */
  C_R1_PTS = C_SC + 6, C_R2_PTS, C_M_PTS, C_W_PTS, C_W_PLO, C_W_PHI,
  C_R1_D, C_R2_D = C_R1_D + TS_LIMBS, C_M_D = C_R2_D + TS_LIMBS, C_W_D = C_M_D + TS_LIMBS,
  CPU_WIDTH = C_W_D + TS_LIMBS
};
enum { SC_HALT = 0, SC_WRITE, SC_COMMIT, SC_DEFER, SC_HINT_LEN, SC_HINT_READ };

/* ---- keccak chip: p3-keccak-air's 2633 columns (zksp_oracle.h KA_*) + the call time ---- */
#define KC_TS KA_WIDTH
#define KECCAK_WIDTH (KA_WIDTH + 1)

/* ---- keccak-memory chip: 50 rows per call, one state word each ---- */
enum {
  KM_IS_REAL = 0, KM_TS, KM_PTR_LO, KM_PTR_HI, KM_IDX, KM_ISF, KM_ISL, KM_CALL, KM_ADDR, KM_OLD_LO, KM_OLD_HI, KM_NEW_LO,
  KM_NEW_HI, KM_PTS, KM_D, KMEM_WIDTH = KM_D + TS_LIMBS
};
/* ---- memory boundary chip: every touched address once, strictly increasing ---- */
enum { MF_IS_REAL = 0, MF_ADDR, MF_IS_INIT, MF_FIN_LO, MF_FIN_HI, MF_FIN_TS, MF_DIFF, MF_INIT = MF_DIFF + 32, MEMFINAL_WIDTH = MF_INIT + 32 };
/* ---- image chip: preprocessed (addr, lo, hi), main (used) ---- */
enum { IMG_P_ADDR = 0, IMG_P_LO, IMG_P_HI, IMAGE_PREP_WIDTH };
#define IMAGE_WIDTH 1
/* ---- program chip: preprocessed instruction fields, main (multiplicity) ---- */
enum { PR_PC = 0, PR_OP, PR_WR, PR_USE2, PR_RD, PR_RS1, PR_RS2, PR_IMM_LO, PR_IMM_HI, PR_TGT, PROGRAM_PREP_WIDTH };
#define PROGRAM_WIDTH 1
/* ---- multiplier chip ---- */
enum { MU_IS_REAL = 0, MU_HI, MU_B, MU_C = MU_B + 32, MU_P = MU_C + 32, MU_Q0 = MU_P + 64, MU_Q1 = MU_Q0 + 10, MU_Q2 = MU_Q1 + 11, MUL_WIDTH = MU_Q2 + 10 };

/* ---- range table: preprocessed (value = row index), main (multiplicity); always 2^TS_LIMB_BITS rows ---- */
#define RANGE_PREP_WIDTH 1
#define RANGE_WIDTH 1
#define RANGE_LOG_H TS_LIMB_BITS

/* ---- buses ---- */
enum { BUS_MEM = 1, BUS_PROG, BUS_KCALL, BUS_KIO, BUS_MUL, BUS_PUBC, BUS_PUBH, BUS_RANGE };

/* A linear form over the row [preprocessed | main]: c0 + sum coef[i] * row[col[i]] (canonical words). */
#define LF_MAX 40
typedef struct { int n; int col[LF_MAX]; uint32_t coef[LF_MAX]; uint32_t c0; } orc_lf;
#define INTER_MAX_ELEMS 10
typedef struct { int bus; int sign; /* +1 send / produce, -1 receive / consume */ orc_lf mult; int n_el; orc_lf el[INTER_MAX_ELEMS]; } orc_inter;

typedef struct {
  const char* name;
  int prep_width, main_width;
  int n_inter;
  const orc_inter* inter;
  int n_constraints;  /* base-field constraints; the LogUp constraints follow */
} orc_chip;
const orc_chip* orc_machine_chip(int chip);
static inline int orc_chip_helpers(const orc_chip* c) { return (c->n_inter + 1) / 2; }
static inline int orc_chip_perm_width(const orc_chip* c) { return 4 * (orc_chip_helpers(c) + 1); }

/* ---- inputs: exactly the arrays zksp_mtrace_section() exposes ---- */
typedef struct {
  const uint32_t* program; size_t n_program;   /* 9 u32 per row */
  const uint32_t* image; size_t n_image;       /* 2 u32 per row */
  uint32_t entry, text_base;
  int log_prog, log_image;
  const uint32_t* cycles; size_t n_cycles;     /* 12 u32 per cycle */
  const uint8_t* keccak; size_t n_keccak;      /* 408 bytes per call */
  const uint32_t* memfinal; size_t n_memfinal; /* 5 u32 */
  const uint32_t* muls; size_t n_muls;         /* 3 u32 */
  const uint32_t* prog_mult;                   /* n_program */
  const uint32_t* image_used;                  /* n_image */
} orc_machine_input;

/* log2 trace height of every chip for this input (minimum 5) */
void orc_machine_heights(const orc_machine_input* in, int logh[N_CHIPS]);
/* column-major traces: prep [prep_width][H] (NULL when the chip has none), main [main_width][H] */
void orc_machine_fill(const orc_machine_input* in, int chip, int logh, uint32_t* prep, uint32_t* main_);
/* Public scalars of a CPU instance: pc and time of its first row, whether another instance continues it, and the pc
 * that one starts at (the hand-over pc, a proof-header word the transcript absorbs). */
enum { CPUPUB_START_PC = 0, CPUPUB_START_TS, CPUPUB_HAS_SUCC, CPUPUB_END_PC, CPUPUB_N };
void orc_machine_cpu_pub(const orc_machine_input* in, int chip, uint32_t pub[CPUPUB_N]);
/* base constraints of one chip at one row pair; `pub` = the CPUPUB_* words for the CPU instances (ignored elsewhere) */
void orc_machine_constraints(int chip, const uint32_t* prep, const uint32_t* loc, const uint32_t* nxt, uint32_t is_first,
                             uint32_t is_last, uint32_t is_trans, const uint32_t* pub, uint32_t* out);

/* ---- whole machine proof ---- */
typedef struct {
  uint32_t exit_code;
  uint32_t pv_len;
  uint32_t pv_digest[8];
  uint32_t deferred_digest[8];
} orc_machine_public;
#define ZKSP_VERSION_MACHINE 5u
/* vk: preprocessed commitment root + digest binding entry pc, table heights and keccak mode */
void orc_machine_setup(const orc_machine_input* in, int keccak_mode, uint32_t prep_root[8], uint32_t vk_digest[8]);
size_t orc_machine_proof_size(const int logh[N_CHIPS], int log_prog, int log_image, const orc_config* cfg, uint32_t pv_len);
int orc_machine_prove(const orc_machine_input* in, int keccak_mode, const orc_machine_public* pub, const uint8_t* public_values,
                      const orc_config* cfg, uint8_t* out, size_t cap, size_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
