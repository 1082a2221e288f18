// The constraint identity at zeta as a FIXED straight-line program over extension values (the first
// piece of stage 2c).  The host verifier checks the identity by instantiating the AIR templates of
// air_machine.hpp over extension values (mverifier.cpp ZetaCtx).  Instantiated over a RECORDING value type instead, the same
// templates emit what they compute: a list of operations  c = a * b + d  over numbered cells, the same for every proof of an AIR
// version - what an arithmetic chip will execute row by row (a write-once memory on a LogUp bus for the cells, the list itself
// as preprocessed columns of the table chip).  A proof's heights enter through input cells only (per chip: the selectors at
// zeta, alpha^offset, cum / H, the instance's public scalars; for the final combination: kappa_c, u_c, v_c), never through the
// program's shape.  This file: the recorder, the program, and an interpreter that runs it on a proof's values - with which
// verify_machine_proof cross-checks its native evaluation when asked to (zksp_zeta_program_selftest, tests/test_zeta_program.py).
#pragma once
#include <cstdint>
#include <vector>

#include "machine_defs.hpp"

namespace zksp {

struct ZetaOp {
  uint32_t a, b, d, c;  // cell[c] = cell[a] * cell[b] + cell[d]
};

// Where a chip's inputs live.  Opened values are addressed canonically - (chip, kind, column) - not by their position in a proof.
struct ZetaChipCells {
  uint32_t prep, main, main_next, perm, perm_next_phi, quot;  // first cell of each run of opened values (perm: 4 base columns per slot)
  uint32_t first, trans, last, apow0, cum_step;               // verifier constants of the chip for this proof
  uint32_t pub[mach::kNumCpuPub];                             // the CPU instances' public scalars (zero cells elsewhere)
  uint32_t kappa, u, v;                                       // weights of the final combination
  uint32_t acc;                                               // output: the chip's folded constraints at zeta
};

struct ZetaProgram {
  std::vector<ZetaOp> ops;
  std::vector<uint32_t> const_cell;   // cells holding constants ...
  std::vector<uint32_t> const_monty;  // ... and their values (base-field Montgomery words, embedded)
  uint32_t n_cells = 0;
  uint32_t zero = 0, one = 0, basis[4] = {0, 0, 0, 0};  // constant cells: 0, 1, the extension's basis 1, X, X^2, X^3
  uint32_t alpha = 0, gamma = 0, beta = 0;             // challenge input cells
  ZetaChipCells chip[mach::kNumChips];
  uint32_t result = 0;  // output: sum_c kappa_c acc_c - sum_c (u_c Q0_c - v_c Q1_c), zero for a proof whose identity holds
  uint32_t n_inputs = 0;
  size_t ops_of_chip[mach::kNumChips] = {0};
  // how often each cell is read (as a, b or d of an operation; the chips' acc cells and the result once more: by whoever
  // checks them): the multiplicity with which the cell's one writer puts it on the arithmetic chip's memory bus - a property
  // of the program, not of a proof
  std::vector<uint32_t> reads;
  uint32_t max_reads = 0, inputs_read = 0;
};

// built once per process (some 10^5 operations)
const ZetaProgram& zeta_program();
// cells: n_cells extension values with the inputs filled in (constants are filled here); runs every operation
void zeta_program_run(const ZetaProgram& zp, Fp4* cells);
// The arithmetic chip's memory argument on the values of a run, natively: every cell is written once - by an operation, or from
// outside (inputs, constants) - with its read count as multiplicity, and read by the operations that use it;
// sum reads[c] / (gamma + fp(c, cell)) over the writes - sum 1 / (gamma + fp(.)) over the reads must vanish (fp: the address and
// the four words against powers of beta).  True if it does.
bool zeta_program_memory_balances(const ZetaProgram& zp, const Fp4* cells, const Fp4& gamma, const Fp4& beta);

}  // namespace zksp
