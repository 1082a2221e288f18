// Launchers of the machine-proof kernels: trace expansion of every
// chip from the executor's records, the generic LogUp permutation trace, per-chip quotient
// evaluation, the mixed-height Merkle commitment, reduced openings per height, proof assembly.
// Same conventions as kernels.h: everything is enqueued on `stream`, nothing allocates or
// synchronises, field elements in device buffers are Montgomery residues, the leading index of
// every per-proof buffer is the proof in the batch.
#pragma once
#include "../host/machine_defs.hpp"
#include "kernels.h"

namespace zksp {

// One matrix of a batch: column-major [width][rows] per proof, proofs `bstride` words apart
// (bstride = 0: one matrix shared by the whole batch, e.g. the preprocessed tables).
struct Seg {
  const uint32_t* p;
  size_t bstride;
  int width;
};
constexpr int kMaxSegs = 27;  // at least kNumChips: every chip could have the same height
static_assert(kMaxSegs >= mach::kNumChips, "a height group may hold every chip");

// ---- trace expansion (row a3 of the machine proof) ----
struct MachineRecords {
  const uint32_t* cycles;      // [B][cap_cycles][12]
  const uint8_t* kcalls;       // [B][cap_keccak][408]
  const uint32_t* memfinal;    // [B][cap_memfinal][5]; row 0 is x0, closed at its last real access (the padding rows move that)
  const uint32_t* muls;        // [B][cap_muls][3]
  const uint32_t* alu_idx;     // [B][cap_alu]: cycle index of every ALU-chip row
  const uint32_t* sub_idx;     // [B][cap_sub]: cycle index of every sub-word-chip row
  const uint32_t* bw_idx;      // [B][cap_bw]: cycle index of every bitwise-chip row
  const uint32_t* ecall_idx;   // [B][cap_ecall]: cycle index of every ecall
  const uint32_t* div_idx;     // [B][cap_div]: cycle index of every divider-chip row (div divu rem remu)
  const uint32_t* agg_heap;    // [B][cap_agg][kP2RecWords]: one record per row of the Poseidon2 chip (air_machine.hpp): flags, tag,
                               //          key, mask, the 16 input words (canonical) - heap nodes of an aggregation payload, then
                               //          the sponges and path steps of a leaf-proof check
  const uint32_t* fold_rows;   // [B][cap_fold][kQrRecWords]: one record per row of the query chip (the row itself)
  const uint32_t* tr_rows;     // [B][cap_tr][kTrRecWords]: one record per row of the transcript chip
  const P2Consts* consts;      // Poseidon2 constants (the Poseidon2 chip's rows are permutations)
  const uint32_t* prog_mult;   // [B][2^log_prog]; the padding row (n_program - 1) holds 0: its fetches follow from cpu_rows
  const uint32_t* counts;      // [B][14]: cycles, keccak calls, memfinal rows, muls, ALU rows, sub-word rows, last time x0 was
                               //          accessed by a real cycle, bitwise rows, Poseidon2-chip rows, ecalls, query-chip rows, divider rows, transcript-chip rows, hint-chip rows
  uint32_t* table_hist;        // [B][kTableWidth][2^16] scratch: multiplicities of the table chip, counted on the device
  uint32_t row0[mach::kNumChips];  // first cycle / event of the chip's instance (second instances: rows of the first)
  size_t cap_cycles, cap_keccak, cap_memfinal, cap_muls, cap_alu, cap_sub, cap_bw, cap_agg, cap_ecall, cap_fold, cap_div, cap_tr;
  const uint32_t* program;     // [n_program][9] (shared); the last row is the padding instruction
  uint32_t text_base, n_program, n_image;
  uint32_t cpu_rows;           // rows of the two CPU instances together: the rows past the last cycle fetch the padding instruction
};
constexpr int kCountWords = 14;
// trace: [B][main_width][2^logh] of the given chip (every chip but kKeccak and kTable)
void launch_machine_trace(hipStream_t stream, int chip, const MachineRecords& rec, uint32_t* trace, int logh, int batch);
// keccak chip: p3-keccak-air's columns by launch_keccak_trace (kernels.h, with a batch stride), then the call time
void launch_keccak_ts(hipStream_t stream, const MachineRecords& rec, uint32_t* trace, size_t trace_bstride, int logh,
                      int batch);
// Table chip multiplicities: what the rows of `chip` (its main trace [B][w][2^logh], Montgomery) look up on the RANGE,
// BYTES and BYTEOP buses is added to rec.table_hist (zero it first: launch_table_clear), by evaluating the chip's own receives -
// so the buses balance by construction whenever every looked-up value has a table row.  launch_table_trace then
// writes the table chip's main columns.
void launch_table_clear(hipStream_t stream, const MachineRecords& rec, int batch);
void launch_cpu_table_count(hipStream_t stream, const uint32_t* trace, int logh, const MachineRecords& rec, int batch);
void launch_table_count(hipStream_t stream, const mach::Interaction* inter, int n_inter, const uint32_t* trace, int width, int logh,
                        const MachineRecords& rec, int batch);
void launch_table_trace(hipStream_t stream, const MachineRecords& rec, uint32_t* trace, int batch);

// ---- mixed-height Merkle commitment (row a5) ----
// Leaf digests of one height group: rows of `nseg` LDE matrices ([w][2H] each) concatenated, written to
// digests[b][c * H + bitrev(m)] (8 words each; `out_bstride` words between proofs).
void launch_mmcs_leaves(hipStream_t stream, const Seg* segs, int nseg, int logh, uint32_t* digests, size_t out_bstride,
                        int batch, const P2Consts* consts);
// One level: out[i] = compress(in[2i], in[2i+1]), then compress(out[i], inject[i]) when inject != null.
// the last levels of a tree (at most 256 nodes each) in one launch; offsets in words from the proof's tree
constexpr int kMmcsTopLevels = 9, kMmcsTopNodes = 256;
struct MmcsTopArgs {
  uint32_t* tree;
  size_t tree_bstride;
  int n_levels;
  int count[kMmcsTopLevels];
  size_t in_off[kMmcsTopLevels], out_off[kMmcsTopLevels], inj_bstride[kMmcsTopLevels];
  const uint32_t* inject[kMmcsTopLevels];
};
// subtrees = 1: the last levels (at most 256 nodes each); subtrees = S: levels whose node counts are multiples of S
void launch_mmcs_top(hipStream_t stream, const MmcsTopArgs& a, int subtrees, int batch, const P2Consts* consts);
void launch_mmcs_level(hipStream_t stream, const uint32_t* in, size_t in_bstride, uint32_t* out, size_t out_bstride,
                       const uint32_t* inject, size_t inject_bstride, size_t count, int batch, const P2Consts* consts);

// ---- LogUp (row a6, lookup argument) ----
struct PermArgs {
  int chip;
  const mach::Interaction* inter;  // device copy of the chip's interactions
  int n_inter;
  Seg prep, main_;                 // traces [w][H]
  const uint32_t* bus_ch;          // [B][8]: gamma, beta
  const uint32_t* bpow;            // [B][kInterMaxElems + 1] Fp4: powers of beta
  uint32_t* perm;                  // [B][perm_width][H]: the helper columns, then the running sum phi
  int perm_width;
  size_t perm_bstride;
  uint32_t* rowsum;                // [B][H] Fp4 scratch
  uint32_t* slice_sums;            // [B][H / 4096] Fp4 scratch (tall chips: the running sum is scanned in slices)
  uint32_t* cum;                   // [B] Fp4 (this chip's cumulative sum)
  size_t cum_bstride;
  uint32_t h_inv;                  // 1 / H (Montgomery): the running sum steps by rowsum - cum / H, cyclically
  int logh, batch;
  int blk0;                        // (tables of launch_perm_multi: the task's first workgroup in its launch)
};
void launch_perm_trace(hipStream_t stream, const PermArgs& a);
// A small batch's LogUp stage over device tables (every chip needs its own rowsum / slice_sums scratch then): tasks[0] the
// CPU instances, [1] the other chips' terms, [2] the sliced scans, [3] the single-workgroup scans; a chip with many
// interactions takes launch_perm_terms_split by itself, before the scans.
struct PermMulti {
  const PermArgs* tasks[4];
  int n[4], blocks[4];
};
int perm_task_kinds(const PermArgs& a);          // bits: 1 CPU terms, 2 generic terms, 4 split terms, 8 sliced scan, 16 scan
int perm_task_blocks(const PermArgs& a, int kind_bit);
void launch_perm_multi_cpu_terms(hipStream_t stream, const PermMulti& m);
void launch_perm_multi_terms(hipStream_t stream, const PermMulti& m);
void launch_perm_multi_scans(hipStream_t stream, const PermMulti& m);  // after every term kernel
void launch_perm_terms_split(hipStream_t stream, const PermArgs& a);
// public terms of the two verifier-closed buses: out[b] = -(sum over the 16 digest words and the exit code of 1/f)
// per proof: pv digest 8, deferred digest 8, exit code (canonical), then the CpuPub words of the two CPU instances (Montgomery)
constexpr int kPubWords = 17 + mach::kNumCpuInst * mach::kNumCpuPub;
void launch_public_bus(hipStream_t stream, const uint32_t* pub_words /*[B][kPubWords]*/,
                       const uint32_t* bus_ch, const uint32_t* bpow, uint32_t* out, size_t out_bstride, int batch);

// ---- quotient (row a6) ----
struct MQuotArgs {
  int chip;
  const mach::Interaction* inter;
  int n_inter, n_base;          // base constraints; one LogUp constraint per slot follows (machine_defs.hpp "LogUp layout")
  Seg prep, main_, perm;        // LDEs [w][2H]
  const uint32_t* alpha_pows;   // [B][alpha_stride] Fp4
  size_t alpha_bstride;
  const uint32_t* bus_ch;       // [B][8]
  const uint32_t* bpow;         // [B][kInterMaxElems + 1] Fp4
  const uint32_t* cum;          // [B] Fp4, stride cum_bstride
  size_t cum_bstride;
  const uint32_t* tw_fwd;       // [H/2] powers of w_H
  uint32_t shift[2];            // g, g * w_2H  (Montgomery)
  uint32_t zh_inv[2];
  uint32_t wh_inv;
  uint32_t h_inv;               // 1 / H (Montgomery)
  const P2Consts* consts;       // Poseidon2 constants (the Poseidon2 chip's constraints)
  const uint32_t* pubs;         // CPU instances: this instance's CpuPub words per proof (Montgomery), stride pubs_bstride
  size_t pubs_bstride;
  uint32_t* quot;               // [B][8][H]: the quotient of this chip's height (its first chip's buffer)
  int accumulate;               // add to `quot` instead of overwriting (another chip of the same height wrote first)
  uint32_t* partial;            // keccak chip: [B][13][2H] Fp4 scratch; CPU chip: [B][2H] Fp4
  int logh, batch;
};
void launch_machine_quotient(hipStream_t stream, const MQuotArgs& a);

// ---- reduced openings per chip (row a7) ----
struct MReduceArgs {
  Seg mats[4];                 // prep, main, perm, quot LDEs ([w][2H]); width 0 = absent
  const uint32_t* af_pows;     // [B][n_open] Fp4, this chip's first power at `pow_off`
  size_t af_bstride;
  size_t pow_off;
  const uint32_t* opened;      // [B][..] Fp4, this chip's first value at `open_off` (same order as the powers)
  size_t opened_bstride, open_off;
  const uint32_t* zeta;        // [B] Fp4
  const uint32_t* tw_fwd;      // [H/2]
  uint32_t shift[2], w_h;
  uint32_t* partial;           // [B][nchunks][2][2H] Fp4 scratch
  uint32_t* bsum;              // [B][2] Fp4 scratch
  uint32_t* out;               // [B][2H] Fp4: G of this height
  size_t out_bstride;
  int accumulate;              // add to `out` instead of overwriting (another chip of the same height)
  int logh, batch;
};
int mreduce_nchunks(int total_width);
void launch_machine_reduce(hipStream_t stream, const MReduceArgs& a);
// A small batch reduces ALL its openings with two launches over device tables: the sums of alpha^i * opened_i per height,
// then one pass per height over every column of every chip of that height (what the per-chip launches add up one chip
// after the other: the sums are exact, their order does not matter).
struct MRSeg {            // one LDE matrix of one chip
  const uint32_t* p;
  size_t bstride;
  int width;
  int pow1, pow2;         // index of alpha_f's power for column 0 at zeta, and at zeta * w (-1: opened at zeta only)
};
struct MRChip { int open_off, n1, n2; };  // the chip's opened values: n1 at zeta from open_off, then n2 at zeta * w
struct MRHeight {
  int logh, seg0, nseg, chip0, nchips, blk0;
  uint32_t* out;          // [B][2H] Fp4
  size_t out_bstride;
  const uint32_t* tw_fwd;
  uint32_t shift[2], w_h;
};
struct MReduceMulti {
  const MRHeight* heights; int n_heights;
  const MRSeg* segs;
  const MRChip* chips;
  const uint32_t* af_pows; size_t af_bstride;
  const uint32_t* opened; size_t opened_bstride;
  const uint32_t* zeta;
  uint32_t* bsum;         // [B][n_heights][2] Fp4 scratch
  int total_blocks, batch;
};
constexpr int kMReduceMultiPoints = 256;  // LDE points per workgroup: 64 lanes x 4 consecutive points, four such parts share the columns
void launch_machine_reduce_multi(hipStream_t stream, const MReduceMulti& a);
// layer[i] += g[i] (Fp4), n elements per proof
void launch_fri_add(hipStream_t stream, uint32_t* layer, size_t layer_bstride, const uint32_t* g, size_t g_bstride, size_t n,
                    int batch);

// ---- proof assembly ----
struct MRound {
  Seg seg[mach::kNumChips];      // LDE matrix of each chip in the round; width 0 = absent (the kernel's arguments stay below 4 KB)
  int logh[mach::kNumChips];
  const uint32_t* tree;          // [B][(2N - 1) * 8], levels back to back, N = 2 * 2^lm
  size_t tree_bstride;
  int lm;
};
struct MAssembleArgs {
  MRound round[4];
  const uint32_t* cum;         // [B][7] Fp4
  const uint32_t* opened;      // [B][n_open] Fp4
  size_t opened_bstride, n_open;
  const uint32_t* fri_layers;  // as in AssembleArgs
  const uint32_t* fri_trees;
  size_t fri_layer_stride, fri_tree_stride;
  const uint32_t* witness;
  const uint32_t* indices;
  uint32_t* body;
  size_t body_stride;
  int lm, n_queries, batch;
};
static_assert(sizeof(MAssembleArgs) <= 4096, "kernel argument segment");
void launch_machine_assemble(hipStream_t stream, const MAssembleArgs& a);

}  // namespace zksp
