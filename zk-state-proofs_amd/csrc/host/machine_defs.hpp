// Static description of the machine proof's chips for the host (verifier, prover orchestration)
// and, flattened, for the device's generic LogUp kernels: widths, bus interactions as linear forms
// over the row [preprocessed | main], constraint counts.  Column layouts and constraints live in
// ../device/air_machine.hpp.  (sp1-core-machine's chip registry + interaction builder in SP1,
// reference Cargo.lock:7130.)
#pragma once
#include <cstdint>

#include "../device/air_machine.hpp"

namespace zksp {
namespace mach {

constexpr int kLfMax = 16, kInterMaxElems = 13;
struct LinForm {
  int32_t n;
  int32_t col[kLfMax];
  uint32_t coef[kLfMax];  // Montgomery words
  uint32_t c0;            // Montgomery word
};
struct Interaction {
  int32_t bus, sign, n_el;  // sign +1: send / produce, -1: receive / consume
  LinForm mult;
  LinForm el[kInterMaxElems];
};
struct ChipDef {
  const char* name;
  int prep_w, main_w, n_inter;
  const Interaction* inter;
  int n_constraints;
  // The last n_merged interactions are sends with boolean, mutually exclusive multiplicities (one instruction class
  // each): they share one fraction, (sum m_k) / (sum m_k f_k + 1 - sum m_k).
  int n_merged;
  // LogUp layout: the interactions before the merged ones two at a time, then the merged group: SLOTS.  Every slot but
  // the last has a helper column (4 base columns) constrained to the slot's value; the last slot's value is
  // phi_next - phi + cum / H - (sum of the helpers) on every row, cyclically (phi: the running sum, phi_0 = 0).
  int slots() const { return (n_inter - n_merged + 1) / 2 + (n_merged ? 1 : 0); }
  int helpers() const { return slots() - 1; }
  int perm_width() const { return 4 * slots(); }
  int total_constraints() const { return n_constraints + slots(); }
};
const ChipDef& chip_def(int chip);

// Quotients.  The chips of one height share ONE quotient: chip c's constraints are folded with the powers
// alpha^(off_c + k), off_c = the sum of total_constraints() over the chips c' < c of the same height, and the sum over the
// height's chips is divided by the vanishing polynomial once.  The first chip of a height (its "leader") carries the
// quotient's 8 columns (two chunks of one extension element each); the others have none.
inline int quot_leader(const int* logh, int c) {
  for (int c2 = 0; c2 < c; ++c2)
    if (logh[c2] == logh[c]) return c2;
  return c;
}
inline int quot_width(const int* logh, int c) { return quot_leader(logh, c) == c ? 8 : 0; }
inline int quot_alpha_offset(const int* logh, int c) {
  int off = 0;
  for (int c2 = 0; c2 < c; ++c2)
    if (logh[c2] == logh[c]) off += chip_def(c2).total_constraints();
  return off;
}

// magic, version, heights, exit code, pv length, 3 digests, hand-over pc; aggregation payload: leaf count, root, digest of the
// leaf list; public bus tuples (the statement of a leaf-proof check): count, digest of the list
constexpr int kHeaderWords = 2 + kNumChips + 2 + 24 + (kNumCpuInst - 1) + 17 + 9;
// A public bus tuple, as the verifier is given it (16 canonical words): bus, 1 if the verifier sends it (0: receives),
// multiplicity, number of elements, the elements (zero padded).
constexpr int kPubTupleWords = 16;
constexpr uint32_t kMachineVersion = 16;

}  // namespace mach
}  // namespace zksp
