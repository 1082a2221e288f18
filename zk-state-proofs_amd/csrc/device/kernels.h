// Host-callable launchers for the HIP kernels of the proving hot path
// (SURVEY.md section 8a rows a3-a7).  Every launcher enqueues on `stream` and
// returns immediately; none allocates or synchronises (graph-capturable).
// All field elements in device buffers are Montgomery residues (field.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "poseidon2.hpp"

namespace zksp {

// Device-resident challenger state, one per proof in a batch.
struct DevChallenger {
  uint32_t state[16];
  uint32_t inbuf[8];
  uint32_t outbuf[8];
  int32_t n_in;
  int32_t n_out;
};

// ---- NTT / LDE (row a4) ----
// ncols contiguous columns of H = 2^logh evaluations -> coefficient columns
// (bit-reversed order, optional) and blowup-2 coset-major LDE [ncols][2][H].
// tables: tw_fwd/tw_inv [H] per-stage twiddles, [2^(t-1) + j] = w_{2^t}^(+-j) (DeviceDomain::twc_*); in_scale_br [ntables][H] and
// out_scale_br [2][H] are indexed by the bit-reversed coefficient position; column
// `col` uses input table (col >> scale_sel_shift) & scale_sel_mask.
void launch_lde(hipStream_t stream, const uint32_t* in, uint32_t* coefs_br, uint32_t* out, const uint32_t* tw_fwd,
                const uint32_t* tw_inv, const uint32_t* in_scale_br, int scale_sel_shift, int scale_sel_mask,
                const uint32_t* out_scale_br, int logh, size_t ncols);
int lde_configure();  // one-time kernel attribute setup; returns a hipError_t

// ---- Poseidon2 Merkle commitment (row a5) ----
// mat: [batch][width][n_rows] column-major (proof stride mat_stride words);
// tree: [batch][2*n_rows-1][8], leaves first, root last (stride tree_stride words).
// upper = false hashes the leaf layer only (finish with launch_merkle_upper).
void launch_merkle_commit(hipStream_t stream, const uint32_t* mat, size_t mat_stride, int width, int logn,
                          uint32_t* tree, size_t tree_stride, int batch, const P2Consts* consts, bool upper = true);
void launch_merkle_upper(hipStream_t stream, int logn, uint32_t* tree, size_t tree_stride, int batch,
                         const P2Consts* consts);
// states: [n][16] -> permuted in place (test hook)
void launch_poseidon2_permute(hipStream_t stream, uint32_t* states, size_t n, const P2Consts* consts);

// ---- keccak chip (rows a3, a6) ----
// states: [batch][max_perms][25] u64; n_perms: [batch]; trace: [batch][2633][H]
void launch_keccak_trace(hipStream_t stream, const uint64_t* states, int max_perms, const uint32_t* n_perms,
                         uint32_t* trace, int logh, int batch);
void launch_keccak_trace_strided(hipStream_t stream, const uint64_t* states, int max_perms, const uint32_t* n_perms,
                                 uint32_t* trace, size_t trace_bstride, int logh, int batch);
struct QuotientArgs {
  const uint32_t* lde;         // [batch][2633][2][H]
  const uint32_t* lde_p;       // [batch][4][2][H] running-sum columns
  const uint32_t* alpha_pows;  // [batch][3185] Fp4
  const uint32_t* bus_ch;      // [batch][2] Fp4: gamma, beta
  const uint32_t* beta_pows;   // [batch][200] Fp4
  const uint32_t* cum_sum;     // [batch] Fp4
  const uint32_t *sel_first, *sel_trans, *sel_last;  // [2][H]
  const uint32_t* zh_inv;      // [2]
  uint32_t* partial;           // [batch][13][2H] Fp4 scratch
  uint32_t* quot;              // [batch][8][H]
  int logh, batch;
};
void launch_keccak_quotient(hipStream_t stream, const QuotientArgs& a);

// ---- LogUp bus (row a6, lookup argument) ----
// io: [batch][8*R] limbs of the public I/O list (input || keccak-f(input)), zero padded
void launch_keccak_io(hipStream_t stream, const uint64_t* states, int max_perms, const uint32_t* n_perms, uint32_t* io,
                      size_t io_stride, int batch);
// trace [batch][2633][H], bus_ch [batch][2] Fp4, beta_pows [batch][200] Fp4 ->
// phi [batch][4][H] (exclusive running sum of export/f), cum_sum [batch] Fp4; terms: [batch][H] Fp4 scratch
void launch_bus_perm_trace(hipStream_t stream, const uint32_t* trace, const uint32_t* bus_ch, const uint32_t* beta_pows,
                           uint32_t* terms, uint32_t* phi, uint32_t* cum_sum, int logh, int batch);

// ---- openings / FRI (row a7) ----
// out[b][i] = (base[b]*base_mul)^e, e = i or bitrev(i); Fp4 each, base_mul a Montgomery base-field word
void launch_ext_powers(hipStream_t stream, const uint32_t* base, size_t base_stride, uint32_t base_mul, uint32_t* out,
                       size_t out_stride, int n, int bitrev_logn, int batch, int centred = 0);
// Barycentric weight tables out[b][4][H] (Fp4, centred words): table t opens a column given by its evaluations on
// sigma_t <w> at zeta[b]: entry i = ((y^H - 1) / H) w^i / (y - w^i), y = zeta[b] / sigma_t.  Table 0: sigma = 1 / sigma_inv[0];
// table 1: table 0 moved by one place (opens at zeta * w); tables 2, 3: sigma = 1 / sigma_inv[1], 1 / sigma_inv[2].
// tw_fwd: [H/2] powers of w; H = 2^logh >= 16.
void launch_bary_weights(hipStream_t stream, const uint32_t* zeta, size_t zeta_stride, const uint32_t sigma_inv[3], const uint32_t* tw_fwd,
                         uint32_t h_inv, uint32_t* out, size_t out_stride, int logh, int batch);
// coefs_br: [batch][ncols][H]; zpow_br: [batch][npoints][H] Fp4: coefficients with powers (both bit-reversed), or
// evaluations with barycentric weights (both in natural order);
// opened[b][pt*pt_stride + col] (Fp4)
void launch_open(hipStream_t stream, const uint32_t* coefs_br, size_t coefs_stride, int ncols, int logh,
                 const uint32_t* zpow_br, size_t zpow_stride, int npoints, uint32_t* opened, size_t opened_stride,
                 size_t pt_stride, int batch);

// The same for tall matrices (log_h >= 12): the coefficient range is split over workgroups; `scratch` holds
// A small batch opens all its matrices with one launch per kind of kernel (launch_open_multi) over a device table of
// tasks.  The caller fills evals .. npts; open_task_plan chooses kind / nsplit / klen / blocks / cblocks as the
// single-matrix launches below do and returns the words of partial sums the task needs; the caller then sorts the tasks
// by kind, assigns `partial`, and numbers the workgroups (blk0 within the kind, cblk0 over the kinds 1..5).
struct OpenTask {
  const uint32_t* evals;
  size_t cstride;
  const uint32_t* table;
  size_t zstride;
  uint32_t* dst;  // the first proof's destination (opened + offset); proofs are opened_stride words apart
  size_t pts;
  uint32_t* partial;
  int ncols, logh, npts, nsplit, klen, kind, blocks, cblocks, blk0, cblk0;
};
size_t open_task_plan(OpenTask* t, int batch);
void launch_open_multi(hipStream_t stream, const OpenTask* tasks, const int first[6], const int count[6], const int blocks[6],
                       int combine_blocks, size_t opened_stride);
// open_tall_scratch_words(ncols, logh, batch) words of partial sums.
size_t open_tall_scratch_words(int ncols, int logh, int batch);
void launch_open_tall(hipStream_t stream, const uint32_t* coefs_br, size_t coefs_stride, int ncols, int logh,
                      const uint32_t* zpow_br, size_t zpow_stride, int npoints, uint32_t* opened, size_t opened_stride,
                      size_t pt_stride, uint32_t* scratch, int batch);

struct ReduceArgs {
  const uint32_t* lde_t;   // [batch][W][2][H]
  const uint32_t* lde_q;   // [batch][8][2][H]
  const uint32_t* lde_p;   // [batch][4][2][H]
  const uint32_t* af_pows; // [batch][2W+16] Fp4
  const uint32_t* opened;  // [batch][2W+16] Fp4 (trace local, trace next, quotient, running sum local, next)
  const uint32_t* zeta;    // [batch] Fp4
  const uint32_t* xs;      // [2][H] domain points
  uint32_t* partial;       // [batch][nchunks][2H] Fp4 scratch
  uint32_t* bsum;          // [batch][5] Fp4 scratch
  uint32_t* out;           // [batch][2][H] Fp4
  size_t opened_stride;    // words between consecutive proofs in `opened`
  size_t out_stride;       // words between consecutive proofs in `out`
  uint32_t w_h;            // generator of the trace subgroup (Montgomery)
  int width;               // W
  int logh;
  int batch;
};
void launch_reduce_openings(hipStream_t stream, const ReduceArgs& a);
int reduce_nchunks(int width);  // column chunks -> size of ReduceArgs::partial

// layer k: in [batch][2][Hk] Fp4 -> out [batch][2][Hk/2] Fp4; beta: [batch] Fp4;
// tw_inv: [H/2] (powers of w_H^-1, H the ORIGINAL height); xinv_c[2] = (shift_k * w_{2Hk}^c)^-1
void launch_fri_fold(hipStream_t stream, const uint32_t* in, size_t in_stride, uint32_t* out, size_t out_stride,
                     const uint32_t* beta, size_t beta_stride, const uint32_t* tw_inv, int tw_shift, uint32_t xinv0,
                     uint32_t xinv1, int loghk, int batch, const uint32_t* join = nullptr, size_t join_stride = 0);
// (`join`: [batch][2][Hk/2] Fp4 added to the folded layer - the machine proof's reduced opening of that height)
// leaves of layer k: (f[c][m], f[c][m+Hk/2]) -> tree [batch][2*Hk-1][8], full tree built
// Commit, transcript and fold of the FRI layers k_start .. logh-1 (each of at most 512 leaves)
// in one launch, one workgroup per proof.  loff_start / toff_start: words into a proof's layer
// buffer / digests into its tree buffer where layer k_start begins; xinv[2k + c] as launch_fri_fold.
struct FriTailArgs {
  uint32_t* layers;
  size_t layer_stride;
  uint32_t* trees;
  size_t tree_stride;
  DevChallenger* ch;
  uint32_t* betas;
  size_t beta_stride;
  const uint32_t* tw_inv;
  uint32_t xinv[48];
  int logh, k_start;
  size_t loff_start, toff_start;
  // machine proofs: the reduced opening that joins the folding after layer k ([B][2^(logh - k)] Fp4, proofs
  // 4 * 2^(logh - k) words apart), or null
  const uint32_t* join[24];
};
constexpr int kFriTailMaxLogLeaves = 9;
void launch_fri_tail(hipStream_t stream, const FriTailArgs& a, int batch, const P2Consts* consts);
void launch_fri_commit(hipStream_t stream, const uint32_t* layer, size_t layer_stride, int loghk, uint32_t* tree,
                       size_t tree_stride, int batch, const P2Consts* consts);

// ---- Fiat-Shamir on the device (row a8, prover side) ----
void launch_ch_init(hipStream_t stream, DevChallenger* ch, const uint32_t* init_obs, int n_obs, int batch,
                    const P2Consts* consts);
// observe `n_obs` words at obs[b*obs_stride ...] (Montgomery), then sample n_ext
// extension elements to out[b*out_stride ...]
void launch_ch_observe_sample(hipStream_t stream, DevChallenger* ch, const uint32_t* obs, size_t obs_stride, int n_obs,
                              uint32_t* out, size_t out_stride, int n_ext, int batch, const P2Consts* consts, int align = 0);
// (align / pad - machine proofs since format v16: a phase of the transcript ends on a block boundary, a pending block is
// zero-filled before the samples are drawn; the query indices start from a fresh squeeze)
// proof-of-work search: smallest w with sample_bits(bits) == 0 after observing w
void launch_ch_grind(hipStream_t stream, DevChallenger* ch, uint32_t* witness, int bits, int batch,
                     const P2Consts* consts, int pad = 0);
// observe witness, then draw n_queries indices of `index_bits` bits
void launch_ch_queries(hipStream_t stream, DevChallenger* ch, const uint32_t* witness, uint32_t* indices,
                       int n_queries, int pow_bits, int index_bits, int batch, const P2Consts* consts, int align = 0);
// out[b][t] = delta[b]^(desc[t] >> 16) * af[b]^(desc[t] & 0xffff) (Fp4), af = chal[b][0..4], delta = chal[b][4..8]: the
// coefficients of the reduced openings of a machine proof (mverifier.hpp machine_reduce_exponents)
void launch_reduce_coefs(hipStream_t stream, const uint32_t* chal, size_t chal_stride, const uint32_t* desc, uint32_t* out,
                         size_t out_stride, int n, int batch);

// ---- proof assembly ----
struct AssembleArgs {
  const uint32_t* lde_t;   // [batch][W][2][H]
  const uint32_t* tree_t;  // [batch][4H-1][8]
  const uint32_t* lde_q;   // [batch][8][2][H]
  const uint32_t* tree_q;
  const uint32_t* lde_p;   // [batch][4][2][H]
  const uint32_t* tree_p;
  const uint32_t* cum_sum; // [batch] Fp4
  const uint32_t* opened;  // [batch][2W+16] Fp4
  const uint32_t* fri_layers;  // [batch][fri_layer_stride]: layers 0..logh-1 back to back, then the final pair
  const uint32_t* fri_trees;   // [batch][fri_tree_stride]: trees 0..logh-1 back to back
  const uint32_t* witness;     // [batch]
  const uint32_t* indices;     // [batch][n_queries]
  uint32_t* body;              // [batch][body_words] canonical u32
  size_t lde_t_stride, tree_t_stride, lde_q_stride, tree_q_stride, lde_p_stride, tree_p_stride, opened_stride,
      fri_layer_stride, fri_tree_stride, body_stride;
  int width, logh, n_queries, batch;
};
void launch_assemble(hipStream_t stream, const AssembleArgs& a);

// instruction-rate probe (kernels_bench.hip)
void launch_rate_kernel(hipStream_t stream, int which, uint32_t* out, int blocks, int iters);
void launch_perm_rate_kernel(hipStream_t stream, uint32_t* out, int blocks, int iters, const P2Consts* consts);

}  // namespace zksp
