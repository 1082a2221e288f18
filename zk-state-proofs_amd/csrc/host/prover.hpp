// Device prover: batch workspace in HBM and the fixed kernel sequence that turns
// keccak-f permutation inputs into proof bodies without a host round trip.
#pragma once
#include "context.hpp"

namespace zksp {

constexpr int kInitObs = 44;  // vk digest 8, log_h, n_perms, exit halves 2, pv digest halves 16, deferred halves 16
constexpr int kTraceWidth = 2633;
constexpr int kNumConstraints = 3182;   // base-field AIR constraints
constexpr int kNumAllConstraints = 3185;  // + 3 extension-valued LogUp constraints
constexpr int kPermWidth = 4;
constexpr int kBusTuple = 200;

struct Workspace {
  int logh = 0;
  int batch = 0;      // allocated capacity
  int n = 0;          // resident batch size
  int max_perms = 0;
  size_t body_words = 0;
  // strides in u32 words
  size_t fri_layer_stride = 0, fri_tree_stride = 0, open_rows_log = 0, io_rows_log = 0;
  // device buffers
  uint64_t* states = nullptr;
  uint32_t *n_perms = nullptr, *init_obs = nullptr;
  uint32_t *trace = nullptr, *coef_t = nullptr, *lde_t = nullptr, *tree_t = nullptr;
  DevChallenger* ch = nullptr;
  uint32_t *alpha = nullptr, *alpha_pows = nullptr;
  // LogUp bus: public I/O limb matrix + its tree, challenges (gamma, beta), beta powers,
  // per-row terms, running-sum columns and their commitment, cumulative sum
  uint32_t *io = nullptr, *tree_io = nullptr, *bus_ch = nullptr, *beta_pows = nullptr, *bus_terms = nullptr;
  uint32_t *phi = nullptr, *coef_p = nullptr, *lde_p = nullptr, *tree_p = nullptr, *cum_sum = nullptr;
  uint32_t *quot = nullptr, *coef_q = nullptr, *lde_q = nullptr, *tree_q = nullptr;
  uint32_t *zeta = nullptr, *zpow = nullptr, *opened = nullptr, *tree_o = nullptr;
  uint32_t *af = nullptr, *af_pows = nullptr, *bsum = nullptr;
  uint32_t *fri_layers = nullptr, *fri_trees = nullptr, *betas = nullptr;
  uint32_t *witness = nullptr, *indices = nullptr, *body = nullptr;
  std::vector<void*> allocs;
  ~Workspace();
};

// (re)allocates the workspace for `batch` proofs of height 2^logh
int workspace_ensure(Context* ctx, int logh, int batch, int max_perms);
// enqueues the whole proving pass over the resident batch
int prove_resident(Context* ctx);

}  // namespace zksp
