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
  int helpers() const { return (n_inter + 1) / 2; }
  int perm_width() const { return 4 * (helpers() + 1); }
  int total_constraints() const { return n_constraints + helpers() + 3; }
};
const ChipDef& chip_def(int chip);

// magic, version, heights, exit code, pv length, 3 digests, hand-over pc; aggregation payload: leaf count, root, digest of the leaf list
constexpr int kHeaderWords = 2 + kNumChips + 2 + 24 + 1 + 17;
constexpr uint32_t kMachineVersion = 6;

}  // namespace mach
}  // namespace zksp
