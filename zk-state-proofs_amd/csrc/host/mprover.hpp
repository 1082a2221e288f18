// Device prover of the machine proof: batch workspace in HBM and the launch sequence that turns
// traced executions into proof bodies ("ZKSP v15") without a host round trip.  See mprover.cpp.
#pragma once
#include <array>
#include <vector>

#include "../device/kernels_machine.h"
#include "context.hpp"
#include "machine.hpp"
#include "mverifier.hpp"

namespace zksp {

// vk digest, heights, exit halves, digest halves, hand-over pc halves, aggregation: leaf count, root, digest of the leaf list;
// public bus tuples: count, digest of the list
// (zero-filled to a block boundary: since format v16 the main root that follows is a block of its own)
constexpr int kMachineInitObs = (8 + mach::kNumChips + 2 + 16 + 16 + 2 * (mach::kNumCpuInst - 1) + 17 + 9 + 7) / 8 * 8;

// Preprocessed tables of one program on the device (built once per verifying key).
struct PrepDevice {
  static constexpr int kMats = 3;   // image, program, table (chip order)
  static int index_of(int chip) { return chip == mach::kImage ? 0 : chip == mach::kProgram ? 1 : chip == mach::kTable ? 2 : -1; }
  int logh[kMats] = {0, 0, 0};
  uint32_t* tr[kMats] = {nullptr, nullptr, nullptr};     // traces (the permutation trace is built from trace rows)
  uint32_t* coef[kMats] = {nullptr, nullptr, nullptr};
  uint32_t* lde[kMats] = {nullptr, nullptr, nullptr};
  uint32_t* tree = nullptr;        // mixed-height tree over the three tables
  uint32_t* inj = nullptr;         // leaf digests of a shorter group (one level at a time, in stream order)
  uint32_t* program = nullptr;     // [n][9] rows for trace expansion
  uint32_t n_program = 0, n_image = 0, text_base = 0, entry = 0;
  int lm = 0;
  std::vector<void*> allocs;
};

struct MachineWorkspace {
  int logh[mach::kNumChips] = {0};
  int batch = 0, n = 0;
  size_t cap_cycles = 0, cap_keccak = 0, cap_memfinal = 0, cap_muls = 0, cap_alu = 0, cap_sub = 0, cap_bw = 0, cap_agg = 0, cap_fold = 0 /* query chip rows */, cap_tr = 0 /* transcript chip rows */;
  const PrepDevice* prep = nullptr;
  // records
  uint32_t *cycles = nullptr, *memfinal = nullptr, *muls = nullptr, *prog_mult = nullptr, *alu_idx = nullptr, *sub_idx = nullptr,
           *bw_idx = nullptr, *ecall_idx = nullptr, *div_idx = nullptr, *agg_heap = nullptr, *fold_rows = nullptr, *tr_rows = nullptr, *counts = nullptr, *table_hist = nullptr;
  uint8_t* kcalls = nullptr;
  uint64_t* kstates = nullptr;
  uint32_t *n_perms = nullptr, *init_obs = nullptr, *pub_words = nullptr;
  // A second set of record buffers: the next batch is uploaded (copy stream) while the current one is
  // being proven, then machine_activate_spare() swaps the sets.
  struct SpareRecords {
    uint32_t *cycles = nullptr, *memfinal = nullptr, *muls = nullptr, *prog_mult = nullptr, *alu_idx = nullptr, *sub_idx = nullptr,
             *bw_idx = nullptr, *ecall_idx = nullptr, *div_idx = nullptr, *agg_heap = nullptr, *fold_rows = nullptr, *tr_rows = nullptr, *counts = nullptr;
    uint8_t* kcalls = nullptr;
    uint64_t* kstates = nullptr;
    uint32_t *n_perms = nullptr, *init_obs = nullptr, *pub_words = nullptr;
    int n = 0;
    // its own shape: a chunk of other chip heights than the resident batch's is uploaded while that batch is proven, and
    // machine_activate_spare lays the arena out for it
    int logh[mach::kNumChips] = {0};
    size_t cap_cycles = 0, cap_keccak = 0, cap_memfinal = 0, cap_muls = 0, cap_alu = 0, cap_sub = 0, cap_bw = 0, cap_agg = 0, cap_fold = 0 /* query chip rows */, cap_tr = 0 /* transcript chip rows */;
    int batch_hint = 0;
    uint32_t idle_chips = 0;
  } spare;
  int rec_slot = 0;  // Context::rec_arena[rec_slot] holds the resident records, the other one the spare set
  // chips without a single real row in the whole resident batch (bit = chip), of those whose padding rows are constant: an idle
  // chip's constraint polynomials vanish identically, so its share of its height's quotient is zero and need not be computed
  uint32_t idle_chips = 0;
  // Host-side staging of a load (counts, transcript words, keccak states, node rows): kept until the next load into the
  // same record set, because an upload into the spare set is not waited for by the host.
  struct LoadStage {
    std::vector<uint32_t> counts, nperms, obs, pubw;
    std::vector<uint64_t> kst;
    std::vector<std::vector<uint32_t>> agg_heaps;
  } stage[2];                     // [0] the resident set's load, [1] the spare set's
  hipEvent_t spare_loaded = nullptr;  // not owned: recorded behind an asynchronous upload into the spare set (or null)
  // per chip: [0] main, [1] permutation, [2] quotient
  struct Mat { uint32_t *tr = nullptr, *coef = nullptr, *lde = nullptr; int w = 0; };
  Mat mat[mach::kNumChips][3];
  uint32_t* zpow[mach::kNumChips] = {nullptr};
  uint32_t* tree[4] = {nullptr, nullptr, nullptr, nullptr};  // rounds 1..3 (0 is the PrepDevice's)
  uint32_t* inj[4][32] = {{nullptr}};                         // leaf digests of the shorter groups, by log LDE size
  uint32_t* G[32] = {nullptr};                                // reduced openings per log height (the tallest lives in fri_layers)
  DevChallenger* ch = nullptr;
  uint32_t *bus_ch = nullptr, *bpow = nullptr, *cum = nullptr, *pubsum = nullptr, *rowsum = nullptr, *slice_sums = nullptr;
  uint32_t *alpha = nullptr, *alpha_pows = nullptr, *zeta = nullptr, *opened = nullptr, *tree_o = nullptr;
  uint32_t* reduce_desc = nullptr;           // [n_open] exponent descriptors of the reduced openings (machine_reduce_exponents)
  std::vector<uint32_t> reduce_desc_host;    // (the upload reads it asynchronously)
  uint32_t *af = nullptr, *af_pows = nullptr, *bsum = nullptr, *kpartial = nullptr, *reduce_scratch = nullptr;
  // scratch of the side streams (batches of at most Context::kSideMaxBatch proofs): [i] belongs to side stream i
  int n_streams = 1;  // 1 + the side streams in use
  uint32_t *side_rowsum[Context::kSideStreams] = {nullptr}, *side_slice_sums[Context::kSideStreams] = {nullptr},
           *side_bsum[Context::kSideStreams] = {nullptr}, *side_reduce_scratch[Context::kSideStreams] = {nullptr};
  uint32_t *fri_layers = nullptr, *fri_trees = nullptr, *betas = nullptr, *witness = nullptr, *indices = nullptr, *body = nullptr;
  size_t n_open = 0, open_off[mach::kNumChips] = {0}, open_rows_log = 0, alpha_stride = 0;
  size_t fri_layer_stride = 0, fri_tree_stride = 0, body_words = 0;
  int lm = 0;
  // a small batch's opening stage as a table of tasks (kernels.h OpenTask): built for `open_tasks_batch` proofs of this layout
  OpenTask* open_tasks = nullptr;
  uint32_t* open_partial = nullptr;
  std::vector<OpenTask> open_tasks_host;
  int open_tasks_batch = -1, open_first[6] = {0}, open_count[6] = {0}, open_blocks[6] = {0}, open_cblocks = 0;
  // ... and its reduced openings height by height (kernels_machine.h MReduceMulti), built with the opening tasks
  MRHeight* mr_heights = nullptr;
  MRSeg* mr_segs = nullptr;
  MRChip* mr_chips = nullptr;
  uint32_t* mr_bsum = nullptr;
  std::vector<MRHeight> mr_heights_host;
  std::vector<MRSeg> mr_segs_host;
  std::vector<MRChip> mr_chips_host;
  int mr_blocks = 0;
  // ... and its LogUp stage (kernels_machine.h PermMulti): four tables of PermArgs, every chip with its own row sums
  PermArgs* perm_tasks = nullptr;
  uint32_t *perm_rowsum_all = nullptr, *perm_slices_all = nullptr;
  std::vector<PermArgs> perm_tasks_host;
  int perm_n[4] = {0}, perm_blocks[4] = {0}, perm_tasks_batch = -1;
  std::vector<void*> allocs;
  ~MachineWorkspace();
};

// uploads the preprocessed tables of `prog` and commits them; the root must equal vk.prep_root
int machine_prep_ensure(Context* ctx, const MachineProgram& prog, const MachineVk& vk, const PrepDevice** out);
// sizes the workspace for `n` traces and uploads their records.  The batch is proven with ONE shape (chip heights):
// `shape` if given (every trace must fit it), else the heights of the element-wise maximum of the traces' counts.
// With `into_spare` the upload goes to the spare record set on the copy stream (the resident batch and a proving pass
// in flight are untouched; the shape may be another one than the resident batch's).  With `loaded` (spare loads only) the call does not
// wait for the copies: it records `loaded` behind them on the copy stream - the caller keeps the traces alive until the
// event has passed - and machine_activate_spare makes the proving stream wait for it.
int machine_load(Context* ctx, const MachineProgram& prog, const MachineVk& vk, const MachineTrace* const* traces, size_t n,
                 bool into_spare = false, const int* shape = nullptr, hipEvent_t loaded = nullptr);
// makes the spare record set the resident batch; if its shape is another one the arena is laid out again (passes already
// enqueued are unaffected: they hold their pointers, and the stream orders what reuses the memory behind them)
int machine_activate_spare(Context* ctx);
// enqueues the whole proving pass over the resident batch
int machine_prove_resident(Context* ctx);
// minimal chip heights of one run / of the runs `counts` covers
void machine_heights(const MachineProgram& prog, const MachineTrace& t, int logh[mach::kNumChips]);
void machine_heights(const MachineProgram& prog, const MachineCounts& counts, int logh[mach::kNumChips]);
bool machine_fits(const MachineTrace& t, const int* logh);
// pc of the first row of the second CPU instance when the first has 2^logh_cpu rows (a proof-header word)
// pc at which CPU instance `inst` (1 .. kNumCpuInst - 1) starts under the chip heights `logh`: the pc of its first cycle, or the
// padding pc when the run has ended before
uint32_t machine_handover_pc(const MachineProgram& prog, const MachineTrace& t, const int* logh, int inst);
// first cycle of CPU instance `inst` (inst = kNumCpuInst: the rows of all instances)
size_t machine_cpu_row0(const int* logh, int inst);

}  // namespace zksp
