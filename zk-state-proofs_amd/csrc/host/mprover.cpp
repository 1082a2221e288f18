// Replaces the body of the reference's `client.prove(&pk, stdin).run()` (prover/src/bin/main.rs:71-74)
// below the executor for the full statement: sp1-prover / sp1-stark's multi-chip commit -> LogUp ->
// quotient -> open -> FRI flow over the CPU, keccak, keccak-memory, memory-boundary, image, program
// and multiplier chips, as one fixed sequence of HIP kernel launches per batch with the Fiat-Shamir
// transcript kept on the device.  Byte-for-byte the oracle's orc_machine_prove (tests).
#include "mprover.hpp"

#include <algorithm>
#include <functional>
#include <cstring>
#include <type_traits>

#include "prover.hpp"

namespace zksp {

using namespace mach;

MachineWorkspace::~MachineWorkspace() {
  for (void* p : allocs)
    if (p) (void)hipFree(p);
}

namespace {

int ceil_log2(size_t v) {
  int l = 0;
  while (((size_t)1 << l) < v) ++l;
  return l;
}
int at_least5(int l) { return l < 5 ? 5 : l; }

template <class T>
bool dalloc(std::vector<void*>* allocs, T** p, size_t count) {
  void* q = nullptr;
  if (hipMalloc(&q, std::max<size_t>(count, 4) * sizeof(T)) != hipSuccess) return false;
  allocs->push_back(q);
  *p = static_cast<T*>(q);
  return true;
}
size_t layer_off(int logn, int layer) { return ((size_t)2 << logn) - ((size_t)2 << (logn - layer)); }

struct RoundMats {
  Seg seg[kNumChips][2];
  int logh[kNumChips];
};

// A stage whose chips are independent, fanned out over the side streams (Context::side) and joined again: begin() makes the
// side streams wait for what the main stream has enqueued so far, lane(i) is the stream of lane i (0: the main stream),
// end() makes the main stream wait for the side streams.  With one lane (large batches) all of it is the main stream and
// nothing is recorded.  Chips go to lanes by HEIGHT (lane_of): the chips of a height share their quotient, their weight
// tables and their reduced opening, so they stay in order on one stream.
struct StageFork {
  Context* ctx;
  int lanes;
  int lane_of_height[32];
  int lane_of_height_q[32];     // the quotient stage's own balance: its kernels cost by constraints, not by cells
  int lane_of_chip[kNumChips];  // for the stages whose chips share nothing: balanced by cells (longest first)
  bool ok = true;
  StageFork(Context* c, int n_lanes, const int* logh) : ctx(c), lanes(n_lanes) {
    for (int& v : lane_of_height) v = 0;
    for (int& v : lane_of_height_q) v = 0;
    for (int& v : lane_of_chip) v = 0;
    if (lanes <= 1) { lanes = 1; return; }
    for (int i = 0; i < Context::kSideStreams && ok; ++i) {
      if (!ctx->side[i] && hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking) != hipSuccess) ok = false;
      if (!ctx->join_ev[i] && hipEventCreateWithFlags(&ctx->join_ev[i], hipEventDisableTiming) != hipSuccess) ok = false;
    }
    if (ok && !ctx->fork_ev && hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming) != hipSuccess) ok = false;
    if (!ok) { (void)hipGetLastError(); lanes = 1; return; }
    // heights and chips to lanes: the heaviest first, each to the lane with the least work so far (cells as the measure)
    size_t load_h[8] = {0}, load_c[8] = {0}, cells_h[32] = {0}, cells_c[kNumChips];
    for (int ch = 0; ch < kNumChips; ++ch) {
      const ChipDef& d = chip_def(ch);
      // (a chip costs its launches too: at a single proof's rate a launch is worth a quarter of a million cells, and the
      // seven chips of the minimum height are seven dependent launches per stage whatever their 32 rows hold)
      cells_c[ch] = ((size_t)(d.main_w + d.perm_width() + 8) << logh[ch]) + ((size_t)1 << 18);
      cells_h[logh[ch]] += cells_c[ch];
    }
    auto lightest = [&](const size_t* load) {
      int best = 0;
      for (int l = 1; l < lanes; ++l)
        if (load[l] < load[best]) best = l;
      return best;
    };
    bool done_h[32] = {false}, done_c[kNumChips] = {false};
    for (int h = 31; h >= 0; --h)  // (the tallest height first: it stays on the main stream, whatever its share)
      if (cells_h[h]) {
        done_h[h] = true;
        lane_of_height[h] = 0;
        load_h[0] += cells_h[h];
        break;
      }
    for (;;) {
      int h = -1;
      for (int k = 0; k < 32; ++k)
        if (cells_h[k] && !done_h[k] && (h < 0 || cells_h[k] > cells_h[h])) h = k;
      if (h < 0) break;
      done_h[h] = true;
      const int l = lightest(load_h);
      lane_of_height[h] = l;
      load_h[l] += cells_h[h];
    }
    {
      // Quotients: a chip of few rows is one workgroup evaluating all its constraints point by point (the Poseidon2 chip's
      // 353 take 210 us over 64 points), a tall one streams its cells; the chips of a height share their quotient and so a
      // lane.  Microseconds of a single proof, measured: 45 + constraints / 2, or cells / 80 000; the keccak chip 335.
      size_t cost_h[32] = {0}, load_q[8] = {0};
      bool done_q[32] = {false};
      for (int ch = 0; ch < kNumChips; ++ch) {
        const ChipDef& d = chip_def(ch);
        const size_t cells = (size_t)(d.prep_w + d.main_w + d.perm_width()) << logh[ch];
        cost_h[logh[ch]] += ch == kKeccak ? std::max<size_t>(335, cells / 17000) : std::max<size_t>(45 + d.total_constraints() / 2, cells / 80000);
      }
      for (;;) {
        int h = -1;
        for (int k = 0; k < 32; ++k)
          if (cost_h[k] && !done_q[k] && (h < 0 || cost_h[k] > cost_h[h])) h = k;
        if (h < 0) break;
        done_q[h] = true;
        const int l = lightest(load_q);
        lane_of_height_q[h] = l;
        load_q[l] += cost_h[h];
      }
    }
    for (;;) {
      int ch = -1;
      for (int k = 0; k < kNumChips; ++k)
        if (!done_c[k] && (ch < 0 || cells_c[k] > cells_c[ch])) ch = k;
      if (ch < 0) break;
      done_c[ch] = true;
      const int l = lightest(load_c);
      lane_of_chip[ch] = l;
      load_c[l] += cells_c[ch];
    }
  }
  hipStream_t lane(int i) const { return i == 0 || lanes == 1 ? ctx->stream : ctx->side[i - 1]; }
  int lane_of(int logh) const { return lanes == 1 ? 0 : lane_of_height[logh]; }
  int lane_of_q(int logh) const { return lanes == 1 ? 0 : lane_of_height_q[logh]; }
  int lane_of_a_chip(int chip) const { return lanes == 1 ? 0 : lane_of_chip[chip]; }
  void begin() const {
    if (lanes == 1) return;
    (void)hipEventRecord(ctx->fork_ev, ctx->stream);
    for (int i = 1; i < lanes; ++i) (void)hipStreamWaitEvent(ctx->side[i - 1], ctx->fork_ev, 0);
  }
  void end() const {
    if (lanes == 1) return;
    for (int i = 1; i < lanes; ++i) {
      (void)hipEventRecord(ctx->join_ev[i - 1], ctx->side[i - 1]);
      (void)hipStreamWaitEvent(ctx->stream, ctx->join_ev[i - 1], 0);
    }
  }
};

// Mixed-height commitment of one round (kernels_machine.h).  tree: [(2N - 1) * 8] words per proof with
// N = 2 * 2^lm; inj[g]: scratch for the leaf digests of the group whose LDE has 2^g rows.  The leaf digests of the height
// groups are independent of one another (each group's go to a buffer of its own): with a fork they are hashed side by
// side, before the levels - which depend on one another - are climbed on the main stream.
// With `joined_lde` the caller has OPENED the fork and put every chip's LDE on the lane of its height: each group's leaves
// then follow their LDEs on that same lane, with no join in between, and the fork is closed here - so that one lane hashes
// (bound by vector-ALU issue) while another still transforms (bound by memory and LDS), in large batches too.
// With `lde` (small batches) the transforms are enqueued HERE, group by group, each group's leaves right behind its own
// transforms, the groups in the order of their sponge chains: a row of the keccak chip is 330 permutations one after the
// other (1.2 ms however many rows there are) and used to start after every other transform of its lane.
// (the chips of a group whose matrices lie side by side in memory - the workspace lays them out so - go in ONE launch)
struct LdeRound {
  const uint32_t* tr[kNumChips];
  uint32_t* lde[kNumChips];
  size_t cols[kNumChips];  // batch x width; 0: the chip has no matrix in this round
  std::function<void(int first_chip, size_t cols, hipStream_t lane)> launch;
};
int mmcs_commit(hipStream_t s, const RoundMats& rm, uint32_t* tree, size_t tree_bstride, uint32_t* const* inj, int batch,
                const P2Consts* kc, Context* span_ctx = nullptr, const char* leaf_span = nullptr, const StageFork* fork = nullptr,
                bool joined_lde = false, const LdeRound* lde = nullptr) {
  int lm = 0;
  for (int c = 0; c < kNumChips; ++c)
    if (rm.seg[c][0].width) lm = std::max(lm, rm.logh[c]);
  const int logn = lm + 1;
  auto group = [&](int g_logn, Seg* out) {
    int n = 0;
    for (int c = 0; c < kNumChips; ++c)
      for (int k = 0; k < 2; ++k)
        if (rm.seg[c][k].width && rm.logh[c] + 1 == g_logn) out[n++] = rm.seg[c][k];
    return n;
  };
  Seg segs[2 * kNumChips];
  int ns = group(logn, segs);
  const bool side_by_side = fork && fork->lanes > 1;
  bool tallest_done = false;
  if (side_by_side && joined_lde && lde) {
    struct Group { int l; size_t blocks; };
    Group order[32];
    int n_groups = 0;
    for (int l = 0; l < logn; ++l) {
      Seg sg[2 * kNumChips];
      const int n2 = group(logn - l, sg);
      size_t wd = 0;
      for (int i = 0; i < n2; ++i) wd += (size_t)sg[i].width;
      if (n2) order[n_groups++] = Group{l, (wd + 7) / 8};
    }
    std::stable_sort(order, order + n_groups, [](const Group& a, const Group& b) { return a.blocks > b.blocks; });
    for (int k = 0; k < n_groups; ++k) {
      const int l = order[k].l, g_logn = logn - l;
      hipStream_t lane = fork->lane(fork->lane_of(g_logn - 1));
      {
        int chips[kNumChips], nch = 0;
        for (int c = 0; c < kNumChips; ++c)
          if (rm.seg[c][0].width && rm.logh[c] + 1 == g_logn && lde->cols[c]) chips[nch++] = c;
        std::sort(chips, chips + nch, [&](int a, int b) { return lde->tr[a] < lde->tr[b]; });
        const size_t hh = (size_t)1 << (g_logn - 1);
        for (int i = 0; i < nch;) {
          size_t cols = lde->cols[chips[i]];
          int j = i + 1;
          while (j < nch && lde->tr[chips[j]] == lde->tr[chips[i]] + cols * hh && lde->lde[chips[j]] == lde->lde[chips[i]] + cols * 2 * hh)
            cols += lde->cols[chips[j++]];
          lde->launch(chips[i], cols, lane);
          i = j;
        }
      }
      Seg sg[2 * kNumChips];
      const int n2 = group(g_logn, sg);
      if (l == 0) {  // (the tallest height is lane 0, the main stream)
        if (span_ctx && leaf_span) {
          ProfileSpan sp(span_ctx, leaf_span);
          launch_mmcs_leaves(lane, sg, n2, lm, tree, tree_bstride, batch, kc);
        } else {
          launch_mmcs_leaves(lane, sg, n2, lm, tree, tree_bstride, batch, kc);
        }
        tallest_done = true;
      } else {
        launch_mmcs_leaves(lane, sg, n2, g_logn - 1, inj[g_logn], (size_t)8 << g_logn, batch, kc);
      }
    }
  } else if (side_by_side && joined_lde) {
    for (int l = 1; l < logn; ++l) {
      Seg sg[2 * kNumChips];
      const int n2 = group(logn - l, sg);
      if (n2) launch_mmcs_leaves(fork->lane(fork->lane_of(logn - l - 1)), sg, n2, logn - l - 1, inj[logn - l], (size_t)8 << (logn - l), batch, kc);
    }
  } else if (side_by_side) {
    // every group's leaf digests on the lane with the least hashing so far, the groups taken by their permutations (rows x
    // blocks), the heaviest first; the tallest group stays on the main stream (lane 0) and counts as its first load
    size_t load[8] = {0}, work[32] = {0};
    auto blocks = [&](const Seg* sg, int n2) {
      size_t wd = 0;
      for (int i = 0; i < n2; ++i) wd += (size_t)sg[i].width;
      return (wd + 7) / 8;
    };
    load[0] = blocks(segs, ns) << logn;
    for (int l = 1; l < logn; ++l) {
      Seg sg[2 * kNumChips];
      const int n2 = group(logn - l, sg);
      // (a sponge is a chain: a short group of many blocks is bound by its length, not by its rows - count at least 2^14 rows)
      if (n2) work[l] = blocks(sg, n2) << std::max(logn - l, 14);
    }
    fork->begin();
    for (;;) {
      int l = -1;
      for (int k = 1; k < logn; ++k)
        if (work[k] && (l < 0 || work[k] > work[l])) l = k;
      if (l < 0) break;
      int best = 0;
      for (int q = 1; q < fork->lanes; ++q)
        if (load[q] < load[best]) best = q;
      load[best] += work[l];
      work[l] = 0;
      Seg sg[2 * kNumChips];
      const int n2 = group(logn - l, sg);
      launch_mmcs_leaves(fork->lane(best), sg, n2, logn - l - 1, inj[logn - l], (size_t)8 << (logn - l), batch, kc);
    }
  }
  if (tallest_done) {
  } else if (span_ctx && leaf_span) {
    ProfileSpan sp(span_ctx, leaf_span);  // exactly one launch: the leaf layer of the tallest matrices
    launch_mmcs_leaves(s, segs, ns, lm, tree, tree_bstride, batch, kc);
  } else {
    launch_mmcs_leaves(s, segs, ns, lm, tree, tree_bstride, batch, kc);
  }
  if (side_by_side) fork->end();
  // small batches: the levels of 65 536 .. 256 nodes as 256 subtrees in one launch, the rest of the tree in another
  MmcsTopArgs mid, top;
  mid.n_levels = top.n_levels = 0;
  static_assert(kMmcsTopNodes == 256 && kMmcsTopLevels == 9, "levels of 256, 128, ..., 1 nodes");
  auto add_level = [&](MmcsTopArgs& g, int l, size_t count, const uint32_t* injp, size_t inj_bstride) {
    if (g.n_levels == 0) { g.tree = tree; g.tree_bstride = tree_bstride; }
    const int k = g.n_levels++;
    g.count[k] = (int)count;
    g.in_off[k] = layer_off(logn, l - 1) * 8;
    g.out_off[k] = layer_off(logn, l) * 8;
    g.inject[k] = injp;
    g.inj_bstride[k] = inj_bstride;
  };
  for (int l = 1; l <= logn; ++l) {
    const size_t count = (size_t)1 << (logn - l);
    const uint32_t* injp = nullptr;
    size_t inj_bstride = 0;
    ns = l < logn ? group(logn - l, segs) : 0;
    if (ns) {
      inj_bstride = (size_t)8 << (logn - l);
      if (!side_by_side) launch_mmcs_leaves(s, segs, ns, logn - l - 1, inj[logn - l], inj_bstride, batch, kc);
      injp = inj[logn - l];
    }
    if (side_by_side && count < (size_t)kMmcsTopNodes) {
      add_level(top, l, count, injp, inj_bstride);
      continue;
    }
    if (side_by_side && count <= (size_t)kMmcsTopNodes * kMmcsTopNodes) {
      add_level(mid, l, count, injp, inj_bstride);
      if (count == (size_t)kMmcsTopNodes) launch_mmcs_top(s, mid, kMmcsTopNodes, batch, kc);
      continue;
    }
    launch_mmcs_level(s, tree + layer_off(logn, l - 1) * 8, tree_bstride, tree + layer_off(logn, l) * 8, tree_bstride, injp,
                      inj_bstride, count, batch, kc);
  }
  if (top.n_levels) launch_mmcs_top(s, top, 1, batch, kc);
  return lm;
}

}  // namespace

size_t machine_cpu_row0(const int* logh, int inst) {
  size_t r = 0;
  for (int i = 0; i < inst; ++i) r += (size_t)1 << logh[cpu_chip(i)];
  return r;
}
uint32_t machine_handover_pc(const MachineProgram& prog, const MachineTrace& t, const int* logh, int inst) {
  const size_t r0 = machine_cpu_row0(logh, inst);
  return r0 < t.cycles.size() ? t.cycles[r0].pc : prog.pad_pc();
}

void machine_heights(const MachineProgram& prog, const MachineCounts& n, int logh[kNumChips]) {
  {
    // CPU instances of one height, as many as the cycles need; the others all padding at the minimum height
    const int hc = at_least5(ceil_log2((n.cycles + kNumCpuInst - 1) / kNumCpuInst));
    for (int i = 0; i < kNumCpuInst; ++i) logh[cpu_chip(i)] = i == 0 || ((size_t)i << hc) < n.cycles ? hc : 5;
  }
  logh[kAlu] = ceil_log2(split_rows(n.alu));
  logh[kAlu2] = ceil_log2(split_rest_rows(n.alu));
  logh[kSub] = ceil_log2(split_rows(n.sub));
  logh[kSub2] = ceil_log2(split_rest_rows(n.sub));
  logh[kBw] = ceil_log2(split_rows(n.bw));
  logh[kBw2] = ceil_log2(split_rest_rows(n.bw));
  logh[kKeccak] = at_least5(ceil_log2(24 * n.keccak));
  logh[kKmem] = at_least5(ceil_log2(50 * n.keccak));
  logh[kMemFinal] = at_least5(ceil_log2(n.memfinal));
  logh[kImage] = prog.log_image;
  logh[kProgram] = prog.log_prog;
  logh[kMul] = at_least5(ceil_log2(n.muls));
  logh[kTable] = kTableLogH;
  logh[kP2] = at_least5(ceil_log2(n.agg ? n.agg : 1));  // a payload's rows: heap nodes, the permutations of a leaf-proof check
  logh[kEcall] = at_least5(ceil_log2(n.ecall));
  logh[kQr] = at_least5(ceil_log2(n.fold ? n.fold : 1));
  logh[kTr] = at_least5(ceil_log2(n.tr ? n.tr : 1));
  logh[kHint] = at_least5(ceil_log2(n.hint ? n.hint : 1));
  logh[kDiv] = at_least5(ceil_log2(n.div));
}
void machine_heights(const MachineProgram& prog, const MachineTrace& t, int logh[kNumChips]) {
  MachineCounts n;
  n.cover(t);
  machine_heights(prog, n, logh);
}
bool machine_fits(const MachineTrace& t, const int* logh) {
  auto two = [&](int a, int b) { return ((size_t)1 << logh[a]) + ((size_t)1 << logh[b]); };
  auto one = [&](int a) { return (size_t)1 << logh[a]; };
  return t.cycles.size() <= machine_cpu_row0(logh, kNumCpuInst) && t.alu_idx.size() <= two(kAlu, kAlu2) && t.sub_idx.size() <= two(kSub, kSub2) &&
         t.bw_idx.size() <= two(kBw, kBw2) && t.p2_rows() <= one(kP2) && t.qr_rows() <= one(kQr) && t.tr_rows() <= one(kTr) && t.hint_words <= one(kHint) &&
         24 * t.keccak.size() <= one(kKeccak) && 50 * t.keccak.size() <= one(kKmem) && t.memfinal.size() <= one(kMemFinal) &&
         t.muls.size() <= one(kMul) && t.ecall_idx.size() <= one(kEcall) && t.div_idx.size() <= one(kDiv);
}

int machine_prep_ensure(Context* ctx, const MachineProgram& prog, const MachineVk& vk, const PrepDevice** out) {
  std::array<uint32_t, 8> key;
  memcpy(key.data(), vk.digest, 32);
  auto it = ctx->prep.find(key);
  if (it != ctx->prep.end()) { *out = it->second.get(); return 0; }
  std::unique_ptr<PrepDevice> pd(new PrepDevice());
  hipStream_t s = ctx->stream;
  std::vector<uint32_t> tr[PrepDevice::kMats];
  machine_prep_traces(prog, &tr[0], &tr[1], &tr[2]);
  pd->logh[0] = prog.log_image;
  pd->logh[1] = prog.log_prog;
  pd->logh[2] = kTableLogH;
  const int widths[PrepDevice::kMats] = {kImagePrepWidth, kProgramPrepWidth, kTablePrepWidth};
  bool ok = true;
  for (int i = 0; i < PrepDevice::kMats && ok; ++i) {
    const size_t h = (size_t)1 << pd->logh[i];
    for (auto& v : tr[i]) v = Fp::from_canonical(v).v;
    ok = ok && dalloc(&pd->allocs, &pd->tr[i], tr[i].size());
    uint32_t* d_tr = pd->tr[i];
    ok = ok && dalloc(&pd->allocs, &pd->lde[i], (size_t)widths[i] * 2 * h);
    if (!ok) break;
    const DeviceDomain* dom = ctx->domain(pd->logh[i]);
    if (!dom) return 3;
    ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(d_tr, tr[i].data(), tr[i].size() * 4, hipMemcpyHostToDevice, s));
    launch_lde(s, d_tr, nullptr, pd->lde[i], dom->twc_fwd, dom->twc_inv, dom->in_scale_br, 0, 0, dom->out_scale_br,
               pd->logh[i], (size_t)widths[i]);
    ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(s));  // tr[i] goes out of use
  }
  pd->lm = std::max(std::max(pd->logh[0], pd->logh[1]), pd->logh[2]);
  const size_t N = (size_t)2 << pd->lm;
  ok = ok && dalloc(&pd->allocs, &pd->tree, (2 * N - 1) * 8);
  ok = ok && dalloc(&pd->allocs, &pd->inj, N * 8);
  ok = ok && dalloc(&pd->allocs, &pd->program, prog.rows.size() * 9);
  if (!ok) return ctx->fail(3, "machine: allocation of the preprocessed tables failed");
  RoundMats rm;
  memset(&rm, 0, sizeof rm);
  rm.seg[kImage][0] = Seg{pd->lde[0], 0, kImagePrepWidth};
  rm.seg[kProgram][0] = Seg{pd->lde[1], 0, kProgramPrepWidth};
  rm.seg[kTable][0] = Seg{pd->lde[2], 0, kTablePrepWidth};
  rm.logh[kImage] = pd->logh[0];
  rm.logh[kProgram] = pd->logh[1];
  rm.logh[kTable] = pd->logh[2];
  uint32_t* inj[32];
  for (auto& p : inj) p = pd->inj;  // a level's injected digests are consumed before the next level writes its own
  mmcs_commit(s, rm, pd->tree, 0, inj, 1, ctx->d_consts);
  static_assert(sizeof(ProgramRow) == 36, "program row layout");
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(pd->program, prog.rows.data(), prog.rows.size() * 36, hipMemcpyHostToDevice, s));
  uint32_t root[8];
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(root, pd->tree + (2 * N - 2) * 8, 32, hipMemcpyDeviceToHost, s));
  ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(s));
  ZKSP_HIP_CHECK(ctx, hipGetLastError());
  for (int i = 0; i < 8; ++i)
    if (Fp::raw(root[i]).to_canonical() != vk.prep_root[i])
      return ctx->fail(3, "machine: the device commitment of the preprocessed tables differs from the verifying key");
  pd->n_program = (uint32_t)prog.rows.size();
  pd->n_image = (uint32_t)prog.image.size();
  pd->text_base = prog.text_base;
  pd->entry = prog.entry;
  *out = pd.get();
  ctx->prep.emplace(key, std::move(pd));
  return 0;
}

// The executor's records of `B` runs with the capacities `t` already holds (cap_*), placed from `base` on (null: sizing
// only).  T: MachineWorkspace (the resident set) or its SpareRecords - the same member names.  Returns the bytes.
template <class T>
static size_t records_place(T* t, void* base, size_t B, const int* logh) {
  size_t off = 0;
  auto A = [&](auto** p, size_t count) {
    using E = std::remove_pointer_t<std::remove_reference_t<decltype(*p)>>;
    const size_t bytes = (std::max<size_t>(count, 4) * sizeof(E) + 255) & ~(size_t)255;
    if (base) *p = reinterpret_cast<E*>(static_cast<char*>(base) + off);
    off += bytes;
  };
  A(&t->cycles, B * t->cap_cycles * 12);
  A(&t->kcalls, B * t->cap_keccak * 408);
  A(&t->kstates, B * t->cap_keccak * 25);
  A(&t->memfinal, B * t->cap_memfinal * 5);
  A(&t->muls, B * t->cap_muls * 3);
  A(&t->alu_idx, B * t->cap_alu);
  A(&t->sub_idx, B * t->cap_sub);
  A(&t->bw_idx, B * t->cap_bw);
  A(&t->ecall_idx, B << logh[kEcall]);  // (an ecall list is as long as the ecall chip is tall, at most)
  A(&t->div_idx, B << logh[kDiv]);
  A(&t->agg_heap, B * t->cap_agg * kP2RecWords);
  A(&t->fold_rows, B * t->cap_fold * kQrRecWords);
  A(&t->tr_rows, B * t->cap_tr * kTrRecWords);
  A(&t->prog_mult, B << logh[kProgram]);
  A(&t->counts, B * kCountWords);
  A(&t->n_perms, B);
  A(&t->init_obs, B * kMachineInitObs);
  A(&t->pub_words, B * kPubWords);
  return off;
}
// rec_arena[slot] of at least `bytes` (grown by a quarter more than asked for; growing frees, which waits for the device)
static int records_ensure(Context* ctx, int slot, size_t bytes) {
  if (ctx->rec_bytes[slot] >= bytes) return 0;
  if (ctx->rec_arena[slot]) (void)hipFree(ctx->rec_arena[slot]);
  ctx->rec_arena[slot] = nullptr;
  ctx->rec_bytes[slot] = 0;
  const size_t want = bytes + bytes / 4;
  if (hipMalloc(&ctx->rec_arena[slot], want) != hipSuccess) {
    (void)hipGetLastError();
    if (hipMalloc(&ctx->rec_arena[slot], bytes) != hipSuccess) return ctx->fail(3, "machine records: hipMalloc of " + std::to_string(bytes >> 20) + " MiB failed");
    ctx->rec_bytes[slot] = bytes;
    return 0;
  }
  ctx->rec_bytes[slot] = want;
  return 0;
}

// `in_flight`: a pass enqueued from the current workspace may still be running (prove_batch going from one shape to the
// next).  The arena is then laid out again WITHOUT waiting for the stream - kernels hold their pointers, and everything that
// will reuse the memory is enqueued behind them - unless it has to grow; the old workspace object (host-side tables an
// asynchronous copy may still read) is retired, not destroyed.
static int workspace_ensure(Context* ctx, const int* logh, int batch, size_t cap_cycles, size_t cap_keccak, size_t cap_memfinal,
                            size_t cap_muls, size_t cap_alu, size_t cap_sub, size_t cap_bw, size_t cap_agg, size_t cap_fold, size_t cap_tr,
                            bool in_flight = false) {
  MachineWorkspace* w = ctx->mws.get();
  if (w && memcmp(w->logh, logh, sizeof w->logh) == 0 && w->batch >= batch && w->cap_cycles >= cap_cycles &&
      w->cap_keccak >= cap_keccak && w->cap_memfinal >= cap_memfinal && w->cap_muls >= cap_muls && w->cap_alu >= cap_alu &&
      w->cap_sub >= cap_sub && w->cap_bw >= cap_bw && w->cap_agg >= cap_agg && w->cap_fold >= cap_fold && w->cap_tr >= cap_tr)
    return 0;
  if (!in_flight) {
    ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->retired.clear();
  }
  {
    std::unique_ptr<MachineWorkspace> fresh(new MachineWorkspace());
    if (w) {
      fresh->rec_slot = w->rec_slot;
      for (int k = 0; k < 2; ++k) fresh->stage[k] = std::move(w->stage[k]);  // (uploads in flight read these vectors' buffers)
      if (in_flight) ctx->retired.push_back(std::move(ctx->mws));
    }
    ctx->mws = std::move(fresh);
  }
  w = ctx->mws.get();
  memcpy(w->logh, logh, sizeof w->logh);
  w->batch = batch;
  w->cap_cycles = cap_cycles; w->cap_keccak = std::max<size_t>(cap_keccak, 1); w->cap_memfinal = cap_memfinal;
  w->cap_muls = std::max<size_t>(cap_muls, 1);
  w->cap_alu = std::max<size_t>(cap_alu, 1); w->cap_sub = std::max<size_t>(cap_sub, 1); w->cap_bw = std::max<size_t>(cap_bw, 1);
  w->cap_agg = std::max<size_t>(cap_agg, 2);
  w->cap_fold = std::max<size_t>(cap_fold, 2);
  w->cap_tr = std::max<size_t>(cap_tr, 2);
  const size_t B = (size_t)batch;
  const uint32_t Q = ctx->params.num_queries;
  // The workspace lives in ONE device arena that survives re-shaping: a batch of other chip heights only lays the same
  // memory out again (pass 0 sizes it, pass 1 hands out the pointers), instead of freeing and re-allocating tens of
  // gigabytes per shape - a block of receipts (BASELINE config 4) goes through a dozen shapes in one prove_batch call.
  bool ok = true;
  size_t arena_need = 0;
  for (int pass = 0; pass < 2 && ok; ++pass) {
    size_t off = 0;
    auto A = [&](auto** p, size_t count) {
      using T = std::remove_pointer_t<std::remove_reference_t<decltype(*p)>>;
      const size_t bytes = (std::max<size_t>(count, 4) * sizeof(T) + 255) & ~(size_t)255;
      if (pass == 1) *p = reinterpret_cast<T*>(static_cast<char*>(ctx->arena) + off);
      off += bytes;
    };
    // (the executor's records are not in this arena: Context::rec_arena, records_place below)
    A(&w->table_hist, (B * kTableWidth) << kTableLogH);
    // The matrices of a round, the traces and the LDEs each, height by height: the columns of the chips of one height lie
    // side by side, so that a small batch transforms them in ONE launch (six CPU instances one after the other were a
    // millisecond of a single proof).  (No coefficient arrays: columns are opened from their evaluations.)
    {
      int order[kNumChips];
      for (int c = 0; c < kNumChips; ++c) order[c] = c;
      std::stable_sort(order, order + kNumChips, [&](int a, int b) { return logh[a] > logh[b]; });
      for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < kNumChips; ++k) {
          const int c = order[k];
          const int wd = r == 0 ? chip_def(c).main_w : r == 1 ? chip_def(c).perm_width() : quot_width(logh, c);  // (one quotient per height: its first chip's)
          w->mat[c][r].w = wd;
          A(&w->mat[c][r].tr, (B * wd) << logh[c]);
        }
        for (int k = 0; k < kNumChips; ++k) {
          const int c = order[k];
          A(&w->mat[c][r].lde, (B * w->mat[c][r].w * 2) << logh[c]);
        }
      }
    }
    int lm = 0;
    size_t n_open = 0, max_total = 0, max_h = 0;
    for (int c = 0; c < kNumChips; ++c) {
      const ChipDef& d = chip_def(c);
      const size_t h = (size_t)1 << logh[c];
      lm = std::max(lm, logh[c]);
      max_h = std::max(max_h, h);
      // barycentric weights: at zeta, at zeta * w_H, and for the two quotient cosets; one set per height
      int first = c;
      for (int c2 = c - 1; c2 >= 0; --c2)
        if (logh[c2] == logh[c]) first = c2;
      if (first == c) A(&w->zpow[c], B * 4 * h * 4);
      else w->zpow[c] = w->zpow[first];
      w->open_off[c] = n_open;
      n_open += (size_t)d.prep_w + 2 * (size_t)d.main_w + 2 * (size_t)d.perm_width() + (size_t)quot_width(logh, c);
      max_total = std::max<size_t>(max_total, (size_t)quot_alpha_offset(logh, c) + (size_t)d.total_constraints());
    }
    w->lm = lm;
    w->n_open = n_open;
    w->alpha_stride = max_total * 4;
    const size_t N = (size_t)2 << lm;
    // (one buffer per distinct height, in both passes alike: the sizing pass cannot test pointers it has not assigned)
    bool seen_inj[4][32] = {{false}}, seen_g[32] = {false};
    for (int r = 1; r < 4; ++r) {
      A(&w->tree[r], B * (2 * N - 1) * 8);
      for (int c = 0; c < kNumChips; ++c)
        if (logh[c] < lm && !seen_inj[r][logh[c] + 1]) {
          seen_inj[r][logh[c] + 1] = true;
          A(&w->inj[r][logh[c] + 1], (B * 8) << (logh[c] + 1));
        }
    }
    for (int c = 0; c < kNumChips; ++c)
      if (logh[c] < lm && !seen_g[logh[c]]) {
        seen_g[logh[c]] = true;
        A(&w->G[logh[c]], (B * 2 * 4) << logh[c]);
      }
    A(&w->ch, B);
    A(&w->bus_ch, B * 8);
    A(&w->bpow, B * (kInterMaxElems + 1) * 4);
    A(&w->cum, B * kNumChips * 4);
    A(&w->pubsum, B * 4);
    A(&w->rowsum, B * max_h * 4);
    A(&w->slice_sums, B * (max_h / 4096 + 1) * 4);
    // the side lanes' scratch, for as many proofs as a pass on side lanes has at most (any workspace can prove a small batch)
    w->n_streams = 1 + Context::kSideStreams;
    const size_t Bs = std::min<size_t>(B, Context::kSideMaxBatch);
    for (int i = 0; i + 1 < w->n_streams; ++i) {
      A(&w->side_rowsum[i], Bs * max_h * 4);
      A(&w->side_slice_sums[i], Bs * (max_h / 4096 + 1) * 4);
      A(&w->side_bsum[i], Bs * 2 * 4);
    }
    A(&w->alpha, B * 4);
    A(&w->alpha_pows, B * w->alpha_stride);
    A(&w->zeta, B * 4);
    w->open_rows_log = (size_t)ceil_log2((n_open * 4 + 7) / 8);
    const size_t R = (size_t)1 << w->open_rows_log;
    A(&w->opened, B * 8 * R);
    A(&w->tree_o, B * (2 * R - 1) * 8);
    A(&w->af, B * 8);  // alpha_f, delta
    A(&w->reduce_desc, n_open);
    A(&w->af_pows, B * n_open * 4);
    A(&w->bsum, B * 2 * 4);
    {
      size_t need = 0;  // partial sums of the reduced openings: [nchunks][2][2H] Fp4 per proof, largest chip
      for (int c = 0; c < kNumChips; ++c) {
        const ChipDef& d = chip_def(c);
        need = std::max(need, (size_t)mreduce_nchunks(d.prep_w + d.main_w + d.perm_width() + quot_width(logh, c)) * 16 * ((size_t)1 << logh[c]));
      }
      for (int c = 0; c < kNumChips; ++c)  // ... and of the tall openings
        for (int wdt : {chip_def(c).prep_w, chip_def(c).main_w, chip_def(c).perm_width(), 8})
          if (wdt) need = std::max(need, open_tall_scratch_words(wdt, logh[c], 1) / 8);  // (words per proof, at eight proofs)
      const size_t nb = std::max<size_t>(B, 8);  // (a batch below eight splits the tall openings finer: sized as for eight)
      A(&w->reduce_scratch, nb * need);
      for (int i = 0; i + 1 < w->n_streams; ++i) A(&w->side_reduce_scratch[i], std::min<size_t>(nb, Context::kSideMaxBatch) * need);
    }
    A(&w->kpartial, B * 13 * (((size_t)2 << logh[kKeccak]) * 4));
    {
      // the task table of a small batch's opening stage and the partial sums of all its tall matrices at once
      size_t words = 0;
      for (int c = 0; c < kNumChips; ++c)
        if (logh[c] >= 12)
          for (int wdt : {chip_def(c).prep_w, chip_def(c).main_w, chip_def(c).perm_width(), quot_width(logh, c) ? 4 : 0, quot_width(logh, c) ? 4 : 0})
            if (wdt) words += open_tall_scratch_words(wdt, logh[c], (int)std::min<size_t>(std::max<size_t>(B, 8), Context::kSideMaxBatch));
      A(&w->open_partial, words);
      A(&w->open_tasks, (size_t)5 * kNumChips);
      A(&w->mr_heights, (size_t)32);
      A(&w->mr_segs, (size_t)4 * kNumChips);
      A(&w->mr_chips, (size_t)kNumChips);
      A(&w->mr_bsum, B * 32 * 2 * 4);
      {
        // the LogUp stage's tables, and row sums / slice sums of every chip at once (batches of at most kSideMaxBatch proofs)
        const size_t Bsm = std::min<size_t>(B, Context::kSideMaxBatch);
        size_t rows = 0, slices = 0;
        for (int c = 0; c < kNumChips; ++c) {
          rows += (size_t)1 << logh[c];
          slices += (((size_t)1 << logh[c]) / 4096 + 1);
        }
        A(&w->perm_tasks, (size_t)4 * kNumChips);
        A(&w->perm_rowsum_all, Bsm * rows * 4);
        A(&w->perm_slices_all, Bsm * slices * 4);
      }
      w->open_tasks_batch = -1;
    }
    w->fri_layer_stride = 0;
    w->fri_tree_stride = 0;
    for (int k = 0; k < lm; ++k) {
      const size_t hk = ((size_t)1 << lm) >> k;
      w->fri_layer_stride += 2 * hk * 4;
      w->fri_tree_stride += (2 * hk - 1) * 8;
    }
    w->fri_layer_stride += 2 * 4;
    A(&w->fri_layers, B * w->fri_layer_stride);
    A(&w->fri_trees, B * w->fri_tree_stride);
    A(&w->betas, B * (size_t)lm * 4);
    A(&w->witness, B);
    A(&w->indices, B * Q);
    w->body_words = machine_proof_body_words(logh, Q);
    A(&w->body, B * w->body_words);
    if (pass == 0) {
      arena_need = off;
      if (ctx->arena_bytes < arena_need) {
        if (in_flight) {  // the memory goes back to the runtime: whatever is enqueued has to be over first
          (void)hipStreamSynchronize(ctx->stream);
          if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
        }
        if (ctx->arena) (void)hipFree(ctx->arena);
        ctx->arena = nullptr;
        ctx->arena_bytes = 0;
        if (hipMalloc(&ctx->arena, arena_need) != hipSuccess) { ok = false; break; }
        ctx->arena_bytes = arena_need;
      }
    }
  }
  if (!ok) {
    ctx->mws.reset();
    return ctx->fail(3, "machine workspace: hipMalloc of " + std::to_string(arena_need >> 20) + " MiB failed");
  }
  ZKSP_HIP_CHECK(ctx, hipMemsetAsync(w->opened, 0, (B * 8 * 4) << w->open_rows_log, ctx->stream));
  machine_reduce_exponents(logh, &w->reduce_desc_host);
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(w->reduce_desc, w->reduce_desc_host.data(), w->reduce_desc_host.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

namespace {
void swap_records(MachineWorkspace* w) {
  MachineWorkspace::SpareRecords& p = w->spare;
  std::swap(w->cycles, p.cycles); std::swap(w->memfinal, p.memfinal); std::swap(w->muls, p.muls);
  std::swap(w->prog_mult, p.prog_mult); std::swap(w->alu_idx, p.alu_idx); std::swap(w->sub_idx, p.sub_idx); std::swap(w->bw_idx, p.bw_idx); std::swap(w->ecall_idx, p.ecall_idx); std::swap(w->div_idx, p.div_idx); std::swap(w->agg_heap, p.agg_heap); std::swap(w->fold_rows, p.fold_rows); std::swap(w->tr_rows, p.tr_rows);
  std::swap(w->counts, p.counts);
  std::swap(w->kcalls, p.kcalls); std::swap(w->kstates, p.kstates); std::swap(w->n_perms, p.n_perms);
  std::swap(w->init_obs, p.init_obs); std::swap(w->pub_words, p.pub_words); std::swap(w->n, p.n);
  std::swap(w->cap_cycles, p.cap_cycles); std::swap(w->cap_keccak, p.cap_keccak); std::swap(w->cap_memfinal, p.cap_memfinal);
  std::swap(w->cap_muls, p.cap_muls); std::swap(w->cap_alu, p.cap_alu); std::swap(w->cap_sub, p.cap_sub); std::swap(w->cap_bw, p.cap_bw);
  std::swap(w->cap_agg, p.cap_agg); std::swap(w->cap_fold, p.cap_fold); std::swap(w->cap_tr, p.cap_tr);
  std::swap(w->idle_chips, p.idle_chips);
  w->rec_slot ^= 1;
}
}  // namespace

int machine_activate_spare(Context* ctx) {
  MachineWorkspace* w = ctx->mws.get();
  if (!w || w->spare.n == 0) return ctx->fail(1, "machine_activate_spare: nothing was loaded into the spare set");
  if (w->spare_loaded) {  // an upload nobody waited for: the passes enqueued from now on do
    ZKSP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, w->spare_loaded, 0));
    w->spare_loaded = nullptr;
  }
  if (memcmp(w->spare.logh, w->logh, sizeof w->logh) != 0 || w->batch < w->spare.n) {
    // another shape (or more runs than the arena is laid out for): lay the arena out again behind the pass in flight.  The
    // spare set is not in the arena and stays where it is; the bodies of the last pass may still be on their way to the
    // host out of the OLD layout, so what is enqueued from now on waits for that copy.
    const MachineWorkspace::SpareRecords sp = w->spare;
    const int slot = w->rec_slot;
    const PrepDevice* prep = w->prep;
    // (the wait goes in BEFORE the new layout's first enqueued work - workspace_ensure ends with a memset of the new `opened`,
    // which may overlap the old layout's `body` that the copy stream is still reading)
    if (ctx->body_free) ZKSP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->body_free, 0));
    int rc = workspace_ensure(ctx, sp.logh, std::max(sp.n, sp.batch_hint), sp.cap_cycles, sp.cap_keccak, sp.cap_memfinal, sp.cap_muls,
                              sp.cap_alu, sp.cap_sub, sp.cap_bw, sp.cap_agg, sp.cap_fold, sp.cap_tr, /*in_flight=*/true);
    if (rc) return rc;
    w = ctx->mws.get();
    w->prep = prep;
    w->rec_slot = slot;
    w->spare = sp;  // (swap_records below makes it the resident set; what it leaves in `spare` is stale and n = 0)
  }
  swap_records(w);
  w->spare.n = 0;
  return 0;
}

int machine_load(Context* ctx, const MachineProgram& prog, const MachineVk& vk, const MachineTrace* const* traces, size_t n,
                 bool into_spare, const int* shape, hipEvent_t loaded) {
  if (n == 0) return ctx->fail(1, "machine_load: empty batch");
  for (size_t i = 0; i < n; ++i)
    if (traces[i]->rec.analysis_fill)
      return ctx->fail(1, "machine_load: a trace was made with ZKSP_UNINIT_FILL set (an analysis aid): unset it to prove");
  const PrepDevice* prep = nullptr;
  int rc = machine_prep_ensure(ctx, prog, vk, &prep);
  if (rc) return rc;
  int logh[kNumChips];
  if (shape) {
    memcpy(logh, shape, sizeof logh);
    if (logh[kImage] != prog.log_image || logh[kProgram] != prog.log_prog || logh[kTable] != kTableLogH)
      return ctx->fail(1, "machine_load: the shape's preprocessed table heights differ from the program's");
  } else {
    MachineCounts cover;
    for (size_t i = 0; i < n; ++i) cover.cover(*traces[i]);
    machine_heights(prog, cover, logh);
  }
  for (size_t i = 0; i < n; ++i)
    if (!machine_fits(*traces[i], logh)) return ctx->fail(1, "machine_load: a trace does not fit the batch's chip heights");
  // record capacities follow from the heights alone, so every batch of these heights fits the same workspace
  const size_t cc = machine_cpu_row0(logh, kNumCpuInst), cm = (size_t)1 << logh[kMemFinal], cu = (size_t)1 << logh[kMul],
               ck = std::min(((size_t)1 << logh[kKeccak]) / 24, ((size_t)1 << logh[kKmem]) / 50),
               ca = ((size_t)1 << logh[kAlu]) + ((size_t)1 << logh[kAlu2]), cs = ((size_t)1 << logh[kSub]) + ((size_t)1 << logh[kSub2]),
               cb = ((size_t)1 << logh[kBw]) + ((size_t)1 << logh[kBw2]),
               cg = (size_t)1 << logh[kP2], cf = (size_t)1 << logh[kQr], ct = (size_t)1 << logh[kTr];  // one record per row of the Poseidon2 / query / transcript chip
  if (logh[kCpu] > 20) return ctx->fail(9, "machine_load: more than 2^21 cycles");
  if (into_spare) {
    MachineWorkspace* w0 = ctx->mws.get();
    if (!w0 || !ctx->copy_stream) return ctx->fail(1, "machine_load: the spare set needs a resident batch and a copy stream");
    // the spare set's own capacities and place (any shape); the buffer it goes to was the resident set two loads ago
    MachineWorkspace::SpareRecords& sp = w0->spare;
    memcpy(sp.logh, logh, sizeof sp.logh);
    sp.cap_cycles = cc; sp.cap_keccak = std::max<size_t>(ck, 1); sp.cap_memfinal = cm; sp.cap_muls = std::max<size_t>(cu, 1);
    sp.cap_alu = std::max<size_t>(ca, 1); sp.cap_sub = std::max<size_t>(cs, 1); sp.cap_bw = std::max<size_t>(cb, 1);
    sp.cap_agg = std::max<size_t>(cg, 2); sp.cap_fold = std::max<size_t>(cf, 2); sp.cap_tr = std::max<size_t>(ct, 2);
    sp.batch_hint = ctx->spare_batch_hint;
    rc = records_ensure(ctx, w0->rec_slot ^ 1, records_place(&sp, nullptr, n, logh));
    if (rc) return rc;
    records_place(&sp, ctx->rec_arena[w0->rec_slot ^ 1], n, logh);
  } else {
    rc = workspace_ensure(ctx, logh, (int)std::max<size_t>(n, (size_t)ctx->batch_hint), cc, ck, cm, cu, ca, cs, cb, cg, cf, ct);
    if (rc) return rc;
    MachineWorkspace* w0 = ctx->mws.get();
    // (the buffer may be in use by a pass that reads the last resident batch: this path is the synchronous one)
    if (ctx->rec_bytes[w0->rec_slot] < records_place(w0, nullptr, n, logh)) ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    rc = records_ensure(ctx, w0->rec_slot, records_place(w0, nullptr, n, logh));
    if (rc) return rc;
    records_place(w0, ctx->rec_arena[w0->rec_slot], n, logh);
    w0->spare.n = 0;
  }
  MachineWorkspace* w = ctx->mws.get();
  w->prep = prep;
  hipStream_t s = into_spare ? ctx->copy_stream : ctx->stream;
  // from here to the end of the function the named record pointers are the set being written
  struct SwapBack {
    MachineWorkspace* w;
    bool on;
    ~SwapBack() { if (on) swap_records(w); }
  } swap_back{w, into_spare};
  if (into_spare) swap_records(w);
  // host staging that outlives the call (an upload into the spare set is not waited for here): the previous load into this
  // record set has long completed
  MachineWorkspace::LoadStage& st = w->stage[w->rec_slot];  // (after the swap above: the buffer being written)
  std::vector<uint32_t>&counts = st.counts, &nperms = st.nperms, &obs = st.obs, &pubw = st.pubw;
  std::vector<uint64_t>& kst = st.kst;
  std::vector<std::vector<uint32_t>>& agg_heaps = st.agg_heaps;
  counts.assign(n * kCountWords, 0); nperms.assign(n, 0); obs.assign(n * kMachineInitObs, 0); pubw.assign(n * kPubWords, 0);
  kst.assign(n * w->cap_keccak * 25, 0);
  agg_heaps.assign(n, std::vector<uint32_t>());
  const size_t hp = (size_t)1 << logh[kProgram];
  uint32_t busy = 0;  // chips (of those that can be idle) with a real row in some run of the batch
  for (size_t i = 0; i < n; ++i) {
    const MachineTrace& t = *traces[i];
    uint32_t* cn = &counts[kCountWords * i];
    cn[0] = (uint32_t)t.cycles.size(); cn[1] = (uint32_t)t.keccak.size();
    cn[2] = (uint32_t)t.memfinal.size(); cn[3] = (uint32_t)t.muls.size();
    cn[4] = (uint32_t)t.alu_idx.size(); cn[5] = (uint32_t)t.sub_idx.size(); cn[6] = t.x0_last; cn[7] = (uint32_t)t.bw_idx.size();
    if (!t.bw_idx.empty())
      ZKSP_HIP_CHECK(ctx, ctx->h2d(w->bw_idx + i * w->cap_bw, t.bw_idx.data(), t.bw_idx.size() * 4, s));
    cn[9] = (uint32_t)t.ecall_idx.size();
    if (!t.ecall_idx.empty())
      ZKSP_HIP_CHECK(ctx, ctx->h2d(w->ecall_idx + (i << logh[kEcall]), t.ecall_idx.data(), t.ecall_idx.size() * 4, s));
    cn[11] = (uint32_t)t.div_idx.size();
    if (!t.div_idx.empty())
      ZKSP_HIP_CHECK(ctx, ctx->h2d(w->div_idx + (i << logh[kDiv]), t.div_idx.data(), t.div_idx.size() * 4, s));
    // aggregation payload: the rows (ancestors of the supplied nodes: key, children's digests) the Poseidon2 chip's
    // rows are expanded from, and its public part
    uint32_t agg_root[8], agg_digest[8];
    const size_t n_agg = t.agg_leaves.size() / 8;
    if (!machine_nodes_public(t.agg_keys.empty() ? nullptr : t.agg_keys.data(), t.agg_leaves.data(), n_agg, agg_root, agg_digest, &agg_heaps[i]))
      return ctx->fail(1, "machine_load: malformed aggregation payload");
    // ... then the rows of a leaf-proof check (the sponges, path steps and injections of its openings; the folds)
    const LeafCheckLog* lc = t.leaf_check.get();
    const size_t n_node = agg_heaps[i].size() / kP2RecWords, n_lc = lc ? lc->p2_rows.size() / kP2RecWords : 0;
    cn[8] = (uint32_t)(n_node + n_lc);
    cn[10] = (uint32_t)t.qr_rows();
    cn[12] = (uint32_t)t.tr_rows();
    cn[13] = (uint32_t)t.hint_words;
    if (cn[8]) busy |= 1u << kP2;
    if (cn[10]) busy |= 1u << kQr;
    if (cn[12]) busy |= 1u << kTr;
    if (!t.div_idx.empty()) busy |= 1u << kDiv;
    if (!t.muls.empty()) busy |= 1u << kMul;
    uint32_t* d_p2 = w->agg_heap + i * w->cap_agg * kP2RecWords;
    if (n_node) ZKSP_HIP_CHECK(ctx, ctx->h2d(d_p2, agg_heaps[i].data(), agg_heaps[i].size() * 4, s));
    if (n_lc) ZKSP_HIP_CHECK(ctx, ctx->h2d(d_p2 + n_node * kP2RecWords, lc->p2_rows.data(), lc->p2_rows.size() * 4, s));
    if (cn[10])
      ZKSP_HIP_CHECK(ctx, ctx->h2d(w->fold_rows + i * w->cap_fold * kQrRecWords, lc->qr_rows.data(), lc->qr_rows.size() * 4, s));
    if (cn[12])
      ZKSP_HIP_CHECK(ctx, ctx->h2d(w->tr_rows + i * w->cap_tr * kTrRecWords, lc->tr_rows.data(), lc->tr_rows.size() * 4, s));
    if (!t.alu_idx.empty())
      ZKSP_HIP_CHECK(ctx, ctx->h2d(w->alu_idx + i * w->cap_alu, t.alu_idx.data(), t.alu_idx.size() * 4, s));
    if (!t.sub_idx.empty())
      ZKSP_HIP_CHECK(ctx, ctx->h2d(w->sub_idx + i * w->cap_sub, t.sub_idx.data(), t.sub_idx.size() * 4, s));
    nperms[i] = (uint32_t)t.keccak.size();
    ZKSP_HIP_CHECK(ctx, ctx->h2d(w->cycles + i * w->cap_cycles * 12, t.cycles.data(), t.cycles.size() * 48, s));
    if (!t.keccak.empty())
      ZKSP_HIP_CHECK(ctx, ctx->h2d(w->kcalls + i * w->cap_keccak * 408, t.keccak.data(), t.keccak.size() * 408, s));
    for (size_t p = 0; p < t.keccak.size(); ++p) memcpy(&kst[(i * w->cap_keccak + p) * 25], t.keccak[p].in, 200);
    ZKSP_HIP_CHECK(ctx, ctx->h2d(w->memfinal + i * w->cap_memfinal * 5, t.memfinal.data(), t.memfinal.size() * 20, s));
    if (!t.muls.empty())
      ZKSP_HIP_CHECK(ctx, ctx->h2d(w->muls + i * w->cap_muls * 3, t.muls.data(), t.muls.size() * 12, s));
    ZKSP_HIP_CHECK(ctx, hipMemsetAsync(w->prog_mult + i * hp, 0, hp * 4, s));
    ZKSP_HIP_CHECK(ctx, ctx->h2d(w->prog_mult + i * hp, t.prog_mult.data(), t.prog_mult.size() * 4, s));
    uint32_t* o = &obs[i * kMachineInitObs];
    memcpy(o, vk.digest, 32);
    for (int c = 0; c < kNumChips; ++c) o[8 + c] = (uint32_t)logh[c];
    o[8 + kNumChips] = t.rec.exit_code & 0xffff;
    o[9 + kNumChips] = t.rec.exit_code >> 16;
    for (int k = 0; k < 8; ++k) {
      o[10 + kNumChips + 2 * k] = t.rec.pv_digest[k] & 0xffff;
      o[11 + kNumChips + 2 * k] = t.rec.pv_digest[k] >> 16;
      o[26 + kNumChips + 2 * k] = t.rec.deferred_digest[k] & 0xffff;
      o[27 + kNumChips + 2 * k] = t.rec.deferred_digest[k] >> 16;
      pubw[i * kPubWords + k] = t.rec.pv_digest[k];
      pubw[i * kPubWords + 8 + k] = t.rec.deferred_digest[k];
    }
    pubw[i * kPubWords + 16] = t.rec.exit_code;
    {
      // the CPU instances: first pc, first time, has a successor, hand-over pc (air_machine.hpp CpuPub)
      constexpr int kHo = 2 * (kNumCpuInst - 1);
      uint32_t start_pc = prog.entry;
      for (int k = 0; k < kNumCpuInst; ++k) {
        uint32_t* cp = &pubw[i * kPubWords + 17 + k * kNumCpuPub];
        const bool succ = k + 1 < kNumCpuInst;
        const uint32_t handover = succ ? machine_handover_pc(prog, t, logh, k + 1) : 0;
        if (succ) {
          o[42 + kNumChips + 2 * k] = handover & 0xffff;
          o[43 + kNumChips + 2 * k] = handover >> 16;
        }
        cp[kPubStartPc] = Fp::from_canonical(start_pc).v;
        cp[kPubStartTs] = Fp::from_canonical((uint32_t)(4 * (machine_cpu_row0(logh, k) + 1))).v;
        cp[kPubHasSucc] = succ ? Fp::one().v : 0u;
        cp[kPubEndPc] = Fp::from_canonical(handover).v;
        cp[kPubPadPc] = Fp::from_canonical(prog.pad_pc()).v;
        start_pc = handover;
      }
      o[42 + kNumChips + kHo] = (uint32_t)n_agg;
      memcpy(o + 43 + kNumChips + kHo, agg_root, 32);
      memcpy(o + 51 + kNumChips + kHo, agg_digest, 32);
      const size_t n_pub = lc ? lc->pub_tuples.size() / kPubTupleWords : 0;
      o[59 + kNumChips + kHo] = (uint32_t)n_pub;
      machine_pub_digest(lc ? lc->pub_tuples.data() : nullptr, n_pub, o + 60 + kNumChips + kHo);
    }
  }
  ZKSP_HIP_CHECK(ctx, ctx->h2d(w->kstates, kst.data(), kst.size() * 8, s));
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(w->counts, counts.data(), counts.size() * 4, hipMemcpyHostToDevice, s));
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(w->n_perms, nperms.data(), nperms.size() * 4, hipMemcpyHostToDevice, s));
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(w->init_obs, obs.data(), obs.size() * 4, hipMemcpyHostToDevice, s));
  ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(w->pub_words, pubw.data(), pubw.size() * 4, hipMemcpyHostToDevice, s));
  if (into_spare && loaded) {
    ZKSP_HIP_CHECK(ctx, hipEventRecord(loaded, s));
    w->spare_loaded = loaded;
  } else {
    ZKSP_HIP_CHECK(ctx, hipStreamSynchronize(s));
    if (into_spare) w->spare_loaded = nullptr;
  }
  w->n = (int)n;
  w->idle_chips = ((1u << kP2) | (1u << kQr) | (1u << kTr) | (1u << kDiv) | (1u << kMul)) & ~busy;
  return 0;
}

static int upload_interactions(Context* ctx) {
  for (int c = 0; c < kNumChips; ++c) {
    if (ctx->d_inter[c]) continue;
    const ChipDef& d = chip_def(c);
    void* p = nullptr;
    ZKSP_HIP_CHECK(ctx, hipMalloc(&p, sizeof(Interaction) * (size_t)d.n_inter));
    ZKSP_HIP_CHECK(ctx, hipMemcpy(p, d.inter, sizeof(Interaction) * (size_t)d.n_inter, hipMemcpyHostToDevice));
    ctx->d_inter[c] = p;
  }
  return 0;
}

int machine_prove_resident(Context* ctx) {
  MachineWorkspace* w = ctx->mws.get();
  if (!w || w->n == 0 || !w->prep) return ctx->fail(1, "machine_prove_resident: no batch loaded");
  int rc = upload_interactions(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const P2Consts* kc = ctx->d_consts;
  const PrepDevice* prep = w->prep;
  const int B = w->n, lm = w->lm;
  const int* logh = w->logh;
  const int Q = (int)ctx->params.num_queries, pow_bits = (int)ctx->params.pow_bits;
  const DeviceDomain* dom[kNumChips];
  for (int c = 0; c < kNumChips; ++c) {
    dom[c] = ctx->domain(logh[c]);
    if (!dom[c]) return 3;
  }
  const size_t N = (size_t)2 << lm, tree_stride = (2 * N - 1) * 8, root_off = (2 * N - 2) * 8;
  auto H = [&](int c) { return (size_t)1 << logh[c]; };
  auto prep_seg = [&](int c, bool lde) -> Seg {
    const int pi = PrepDevice::index_of(c);
    if (pi < 0) return Seg{nullptr, 0, 0};
    return Seg{lde ? prep->lde[pi] : nullptr, 0, chip_def(c).prep_w};
  };

  // ---- main traces ----
  MachineRecords rec;
  rec.cycles = w->cycles; rec.kcalls = w->kcalls; rec.memfinal = w->memfinal; rec.muls = w->muls;
  rec.alu_idx = w->alu_idx; rec.sub_idx = w->sub_idx; rec.bw_idx = w->bw_idx; rec.ecall_idx = w->ecall_idx; rec.div_idx = w->div_idx; rec.agg_heap = w->agg_heap; rec.fold_rows = w->fold_rows; rec.tr_rows = w->tr_rows; rec.consts = kc;
  rec.prog_mult = w->prog_mult; rec.counts = w->counts; rec.table_hist = w->table_hist;
  memset(rec.row0, 0, sizeof rec.row0);
  for (int k = 1; k < kNumCpuInst; ++k) rec.row0[cpu_chip(k)] = (uint32_t)machine_cpu_row0(logh, k);
  rec.row0[kAlu2] = (uint32_t)1 << logh[kAlu]; rec.row0[kSub2] = (uint32_t)1 << logh[kSub]; rec.row0[kBw2] = (uint32_t)1 << logh[kBw];
  rec.cap_cycles = w->cap_cycles; rec.cap_keccak = w->cap_keccak; rec.cap_memfinal = w->cap_memfinal; rec.cap_muls = w->cap_muls;
  rec.cap_alu = w->cap_alu; rec.cap_sub = w->cap_sub; rec.cap_bw = w->cap_bw; rec.cap_agg = w->cap_agg; rec.cap_fold = w->cap_fold; rec.cap_tr = w->cap_tr; rec.cap_ecall = (size_t)1 << logh[kEcall]; rec.cap_div = (size_t)1 << logh[kDiv];
  rec.program = prep->program; rec.text_base = prep->text_base; rec.n_program = prep->n_program; rec.n_image = prep->n_image;
  rec.cpu_rows = (uint32_t)machine_cpu_row0(logh, kNumCpuInst);
  // Small batches: the chips of a stage go to the lanes of a fork (by height) and the stage joins again; large batches have
  // one lane, the main stream, and everything below is enqueued exactly as it reads.
  const StageFork fork(ctx, (int)B <= ctx->lane_max_batch() ? w->n_streams : 1, logh);
  auto SL = [&](int c) { return fork.lane(fork.lane_of(logh[c])); };       // the stream of chip c's height (shared state)
  auto SC = [&](int c) { return fork.lane(fork.lane_of_a_chip(c)); };       // ... of chip c itself (nothing shared)
  auto lane_scratch = [&](int c, uint32_t* main_buf, uint32_t* const* side_bufs) {
    const int l = fork.lane_of(logh[c]);
    return l == 0 ? main_buf : side_bufs[l - 1];
  };
  auto chip_scratch = [&](int c, uint32_t* main_buf, uint32_t* const* side_bufs) {
    const int l = fork.lane_of_a_chip(c);
    return l == 0 ? main_buf : side_bufs[l - 1];
  };
  {
    ProfileSpan sp(ctx, "m_trace");
    // the table chip answers what the others look up: their RANGE / BYTES receives are counted on their finished traces
    launch_table_clear(s, rec, B);
    fork.begin();
    for (int c = 0; c < kNumChips; ++c) {
      if (c == kKeccak) {
        const size_t bs = (size_t)kKeccakWidth * H(c);
        launch_keccak_trace_strided(SC(c), w->kstates, (int)w->cap_keccak, w->n_perms, w->mat[c][0].tr, bs, logh[c], B);
        launch_keccak_ts(SC(c), rec, w->mat[c][0].tr, bs, logh[c], B);
      } else if (c != kTable) {
        launch_machine_trace(SC(c), c, rec, w->mat[c][0].tr, logh[c], B);
      }
      if (is_cpu_chip(c)) launch_cpu_table_count(SC(c), w->mat[c][0].tr, logh[c], rec, B);
      for (int c2 : {(int)kKmem, (int)kMemFinal, (int)kBw, (int)kBw2, (int)kSub, (int)kSub2, (int)kEcall, (int)kP2, (int)kMul, (int)kDiv})
        if (c2 == c)
          launch_table_count(SC(c), static_cast<const Interaction*>(ctx->d_inter[c]), chip_def(c).n_inter, w->mat[c][0].tr,
                             chip_def(c).main_w, logh[c], rec, B);
    }
    fork.end();
    launch_table_trace(s, rec, w->mat[kTable][0].tr, B);
  }
  // A round's LDEs and leaf digests go by height over the lanes of ONE fork, each group's leaves behind its own LDEs with no
  // join in between.  For small batches that is simply more work in flight.  For large ones it was measured as a way to
  // run the hashing of the short chips (bound by vector-ALU issue) beside the transforms of the tall ones (ZKSP_OVERLAP=1:
  // two lanes): 355.1 against 353.9 proofs/s at batch 192 - the chunk transform is bound by butterfly issue itself, the two
  // kinds of kernels share the vector ALUs instead of complementing each other - so large batches keep one lane.
  static const bool overlap = getenv("ZKSP_OVERLAP") != nullptr;
  const StageFork fork2(ctx, (int)B <= ctx->lane_max_batch() ? w->n_streams : (overlap ? 2 : 1), logh);
  auto HL = [&](int c) { return fork2.lane(fork2.lane_of(logh[c])); };
  // small batches: every group's transforms and leaves together, longest sponge first (mmcs_commit, `lde`)
  const bool grouped = fork2.lanes > 1;
  LdeRound lde_main, lde_perm, lde_quot;
  for (int c = 0; c < kNumChips; ++c) {
    LdeRound* rd[3] = {&lde_main, &lde_perm, &lde_quot};
    for (int r = 0; r < 3; ++r) {
      rd[r]->tr[c] = w->mat[c][r].tr;
      rd[r]->lde[c] = w->mat[c][r].lde;
      rd[r]->cols[c] = (size_t)B * w->mat[c][r].w;
    }
  }
  lde_main.launch = [&](int c, size_t cols, hipStream_t lane) {
    launch_lde(lane, w->mat[c][0].tr, nullptr, w->mat[c][0].lde, dom[c]->twc_fwd, dom[c]->twc_inv, dom[c]->in_scale_br, 0, 0,
               dom[c]->out_scale_br, logh[c], cols);
  };
  lde_perm.launch = [&](int c, size_t cols, hipStream_t lane) {
    launch_lde(lane, w->mat[c][1].tr, nullptr, w->mat[c][1].lde, dom[c]->twc_fwd, dom[c]->twc_inv, dom[c]->in_scale_br, 0, 0,
               dom[c]->out_scale_br, logh[c], cols);
  };
  // (quotient: columns 4c..4c+3 of every proof were evaluated over coset c: scale tables 1 and 2)
  lde_quot.launch = [&](int c, size_t cols, hipStream_t lane) {
    if (cols)
      launch_lde(lane, w->mat[c][2].tr, nullptr, w->mat[c][2].lde, dom[c]->twc_fwd, dom[c]->twc_inv, dom[c]->in_scale_br + H(c), 2,
                 1, dom[c]->out_scale_br, logh[c], cols);
  };
  {
    ProfileSpan sp(ctx, "m_lde_main");
    fork2.begin();
    if (!grouped)
      for (int c = 0; c < kNumChips; ++c) lde_main.launch(c, lde_main.cols[c], HL(c));
  }
  RoundMats rm[4];
  memset(rm, 0, sizeof rm);
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < kNumChips; ++c) rm[r].logh[c] = logh[c];
  for (int c = 0; c < kNumChips; ++c) {
    rm[0].seg[c][0] = prep_seg(c, true);
    for (int r = 1; r <= 2; ++r)
      rm[r].seg[c][0] = Seg{w->mat[c][r - 1].lde, (size_t)w->mat[c][r - 1].w * 2 * H(c), w->mat[c][r - 1].w};
    rm[3].seg[c][0] = Seg{w->mat[c][2].lde, (size_t)w->mat[c][2].w * 2 * H(c), w->mat[c][2].w};
  }
  {
    ProfileSpan sp(ctx, "m_commit_main");
    mmcs_commit(s, rm[1], w->tree[1], tree_stride, w->inj[1], B, kc, ctx, "m_leaf_main", &fork2, true, grouped ? &lde_main : nullptr);
  }
  {
    ProfileSpan sp(ctx, "transcript");
    launch_ch_init(s, w->ch, w->init_obs, kMachineInitObs, B, kc);
    launch_ch_observe_sample(s, w->ch, w->tree[1] + root_off, tree_stride, 8, w->bus_ch, 8, 2, B, kc, 1);
    launch_ext_powers(s, w->bus_ch + 4, 8, kR1, w->bpow, (size_t)(kInterMaxElems + 1) * 4, kInterMaxElems + 1, 0, B);
  }
  // ---- LogUp permutation traces ----
  {
    ProfileSpan sp(ctx, "m_perm");
    const bool tabled = fork.lanes > 1;  // small batches: one launch per kind of kernel over tables (launch_perm_multi)
    const bool build = tabled && w->perm_tasks_batch != B;
    std::vector<PermArgs> kinds[4], split;
    size_t row_off = 0, slice_off = 0;
    if (!tabled) fork.begin();
    for (int c = 0; c < kNumChips; ++c) {
      const ChipDef& d = chip_def(c);
      PermArgs pa;
      pa.chip = c;
      pa.inter = static_cast<const Interaction*>(ctx->d_inter[c]);
      pa.n_inter = d.n_inter;
      pa.prep = Seg{d.prep_w ? prep->tr[PrepDevice::index_of(c)] : nullptr, 0, d.prep_w};
      pa.main_ = Seg{w->mat[c][0].tr, (size_t)d.main_w * H(c), d.main_w};
      pa.bus_ch = w->bus_ch;
      pa.bpow = w->bpow;
      pa.perm = w->mat[c][1].tr;
      pa.perm_width = d.perm_width();
      pa.perm_bstride = (size_t)d.perm_width() * H(c);
      pa.h_inv = Fp::from_canonical((uint32_t)(H(c) % kP)).inv().v;
      pa.rowsum = tabled ? w->perm_rowsum_all + row_off : chip_scratch(c, w->rowsum, w->side_rowsum);
      pa.slice_sums = tabled ? w->perm_slices_all + slice_off : chip_scratch(c, w->slice_sums, w->side_slice_sums);
      row_off += (size_t)B * H(c) * 4;
      slice_off += (size_t)B * (H(c) / 4096 + 1) * 4;
      pa.cum = w->cum + 4 * c;
      pa.cum_bstride = (size_t)4 * kNumChips;
      pa.logh = logh[c];
      pa.batch = B;
      pa.blk0 = 0;
      if (!tabled) {
        launch_perm_trace(SC(c), pa);
        continue;
      }
      const int k = perm_task_kinds(pa);
      if (k & 4) split.push_back(pa);  // (a launch of its own; its scan is in the tables)
      if (build) {
        if (k & 1) kinds[0].push_back(pa);
        if (k & 2) kinds[1].push_back(pa);
        if (k & 8) kinds[2].push_back(pa);
        if (k & 16) kinds[3].push_back(pa);
      }
    }
    if (tabled) {
      if (build) {
        static const int bit[4] = {1, 2, 8, 16};
        w->perm_tasks_host.clear();
        for (int q = 0; q < 4; ++q) {
          w->perm_n[q] = (int)kinds[q].size();
          w->perm_blocks[q] = 0;
          for (PermArgs& t : kinds[q]) {
            t.blk0 = w->perm_blocks[q];
            w->perm_blocks[q] += perm_task_blocks(t, bit[q]);
            w->perm_tasks_host.push_back(t);
          }
        }
        ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(w->perm_tasks, w->perm_tasks_host.data(), w->perm_tasks_host.size() * sizeof(PermArgs),
                                           hipMemcpyHostToDevice, s));
        w->perm_tasks_batch = B;
      }
      PermMulti pm;
      const PermArgs* base = w->perm_tasks;
      for (int q = 0; q < 4; ++q) {
        pm.tasks[q] = base;
        pm.n[q] = w->perm_n[q];
        pm.blocks[q] = w->perm_blocks[q];
        base += w->perm_n[q];
      }
      // the CPU instances' terms, the other chips' and the many-interaction chip's side by side, then the scans
      fork.begin();
      launch_perm_multi_cpu_terms(fork.lane(0), pm);
      launch_perm_multi_terms(fork.lane(1 % fork.lanes), pm);
      for (const PermArgs& t : split) launch_perm_terms_split(fork.lane(2 % fork.lanes), t);
      fork.end();
      launch_perm_multi_scans(s, pm);
    }
    if (!tabled) fork.end();
  }
  {
    ProfileSpan sp(ctx, "m_lde_perm");
    fork2.begin();
    if (!grouped)
      for (int c = 0; c < kNumChips; ++c) lde_perm.launch(c, lde_perm.cols[c], HL(c));
  }
  {
    ProfileSpan sp(ctx, "m_commit_perm");
    mmcs_commit(s, rm[2], w->tree[2], tree_stride, w->inj[2], B, kc, nullptr, nullptr, &fork2, true, grouped ? &lde_perm : nullptr);
  }
  {
    ProfileSpan sp(ctx, "transcript");
    launch_ch_observe_sample(s, w->ch, w->tree[2] + root_off, tree_stride, 8, w->alpha, 4, 0, B, kc);
    launch_ch_observe_sample(s, w->ch, w->cum, (size_t)4 * kNumChips, 4 * kNumChips, w->alpha, 4, 1, B, kc, 1);
    launch_ext_powers(s, w->alpha, 4, kR1, w->alpha_pows, w->alpha_stride, (int)(w->alpha_stride / 4), 0, B);
  }
  // ---- quotients ----
  const Fp g = Fp::from_canonical(kGen);
  {
    ProfileSpan sp(ctx, "m_quotient");
    fork.begin();
    for (int c = 0; c < kNumChips; ++c) {
      const ChipDef& d = chip_def(c);
      const size_t h = H(c);
      MQuotArgs qa;
      qa.chip = c;
      qa.inter = static_cast<const Interaction*>(ctx->d_inter[c]);
      qa.n_inter = d.n_inter;
      qa.n_base = d.n_constraints;
      qa.prep = prep_seg(c, true);
      qa.main_ = rm[1].seg[c][0];
      qa.perm = rm[2].seg[c][0];
      qa.alpha_pows = w->alpha_pows + 4 * (size_t)quot_alpha_offset(logh, c);  // the chips of a height share a quotient
      qa.alpha_bstride = w->alpha_stride;
      qa.bus_ch = w->bus_ch;
      qa.bpow = w->bpow;
      qa.cum = w->cum + 4 * c;
      qa.cum_bstride = (size_t)4 * kNumChips;
      qa.tw_fwd = dom[c]->tw_fwd;
      const Fp w2h = fp_root_of_unity(logh[c] + 1), wh = fp_root_of_unity(logh[c]);
      const Fp sh[2] = {g, g * w2h};
      for (int k = 0; k < 2; ++k) {
        qa.shift[k] = sh[k].v;
        qa.zh_inv[k] = (sh[k].pow(h) - Fp::one()).inv().v;
      }
      qa.wh_inv = wh.inv().v;
      qa.h_inv = Fp::from_canonical((uint32_t)(H(c) % kP)).inv().v;
      qa.consts = kc;
      qa.pubs = is_cpu_chip(c) || c == kEcall ? w->pub_words + 17 + (is_cpu_chip(c) ? cpu_instance(c) * kNumCpuPub : 0) : nullptr;  // (ecall chip: the padding pc)
      qa.pubs_bstride = kPubWords;
      qa.quot = w->mat[quot_leader(logh, c)][2].tr;
      qa.accumulate = quot_leader(logh, c) != c;
      if ((w->idle_chips >> c) & 1u) {
        // not a real row in the whole batch and constant padding rows: every constraint polynomial of the chip vanishes
        // identically (a selector times a constant that is zero on the trace domain), and so does its LogUp part (no
        // multiplicity, a zero running sum): nothing to add to the height's quotient.  A single proof spent a tenth of its
        // device time on the 925 constraints of the four chips that only a leaf-proof check or a division gives rows.
        if (!qa.accumulate) ZKSP_HIP_CHECK(ctx, hipMemsetAsync(qa.quot, 0, (size_t)B * 8 * h * 4, fork.lane(fork.lane_of_q(logh[c]))));
        continue;
      }
      // (CPU: 8 H words per proof of the scratch's >= 16 H; the chips of a height share a lane, so its scratch is theirs)
      const int ql = fork.lane_of_q(logh[c]);
      qa.partial = is_cpu_chip(c) ? (ql == 0 ? w->reduce_scratch : w->side_reduce_scratch[ql - 1]) : w->kpartial;
      qa.logh = logh[c];
      qa.batch = B;
      launch_machine_quotient(fork.lane(ql), qa);
    }
    fork.end();
  }
  {
    ProfileSpan sp(ctx, "m_lde_quot");
    fork2.begin();
    if (!grouped)
      for (int c = 0; c < kNumChips; ++c) lde_quot.launch(c, lde_quot.cols[c], HL(c));
  }
  {
    ProfileSpan sp(ctx, "m_commit_quot");
    mmcs_commit(s, rm[3], w->tree[3], tree_stride, w->inj[3], B, kc, nullptr, nullptr, &fork2, true, grouped ? &lde_quot : nullptr);
  }
  const size_t R = (size_t)1 << w->open_rows_log;
  {
    ProfileSpan sp(ctx, "transcript");
    launch_ch_observe_sample(s, w->ch, w->tree[3] + root_off, tree_stride, 8, w->zeta, 4, 1, B, kc);
  }
  // ---- openings at zeta and zeta * w_H ----
  {
    ProfileSpan sp(ctx, "m_open");
    // the weight tables first, one set per height; then every chip against its height's tables
    fork.begin();
    for (int c = 0; c < kNumChips; ++c) {
      bool have = false;  // chips of one height share their tables
      for (int c2 = 0; c2 < c; ++c2) have = have || logh[c2] == logh[c];
      if (have) continue;
      const size_t h = H(c), zs = 4 * h * 4;
      const uint32_t h_inv = Fp::from_canonical((uint32_t)(h % kP)).inv().v;
      const Fp g = Fp::from_canonical(kGen), gw = g * fp_root_of_unity(logh[c] + 1);
      const uint32_t sinv[3] = {kR1, g.inv().v, gw.inv().v};
      launch_bary_weights(SL(c), w->zeta, 4, sinv, dom[c]->tw_fwd, h_inv, w->zpow[c], zs, logh[c], B);
    }
    fork.end();
    const bool tabled = fork.lanes > 1;  // small batches: the whole stage in one launch per kind of kernel
    if (w->open_tasks_batch != B) {  // (the reduced openings' tables serve every batch size)
      std::vector<OpenTask>& tk = w->open_tasks_host;
      tk.clear();
      for (int c = 0; c < kNumChips; ++c) {
        const ChipDef& d = chip_def(c);
        const size_t h = H(c), zs = 4 * h * 4;
        const int pw = d.prep_w, mw = d.main_w, ew = d.perm_width();
        uint32_t* base = w->opened + w->open_off[c] * 4;
        const size_t pt_stride = (size_t)mw + ew + (size_t)w->mat[c][2].w;
        auto add = [&](const uint32_t* evals, size_t cstride, int ncols, int npts, const uint32_t* table, uint32_t* dst, size_t pts) {
          OpenTask t{};
          t.evals = evals; t.cstride = cstride; t.table = table; t.zstride = zs; t.dst = dst; t.pts = pts;
          t.ncols = ncols; t.logh = logh[c]; t.npts = npts;
          tk.push_back(t);
        };
        if (pw) add(prep->tr[PrepDevice::index_of(c)], 0, pw, 1, w->zpow[c], base, 0);
        add(w->mat[c][0].tr, (size_t)mw * h, mw, 2, w->zpow[c], base + (size_t)pw * 4, pt_stride);
        add(w->mat[c][1].tr, (size_t)ew * h, ew, 2, w->zpow[c], base + (size_t)(pw + mw) * 4, pt_stride);
        if (w->mat[c][2].w) {
          add(w->mat[c][2].tr, 8 * h, 4, 1, w->zpow[c] + 2 * h * 4, base + (size_t)(pw + mw + ew) * 4, 0);
          add(w->mat[c][2].tr + 4 * h, 8 * h, 4, 1, w->zpow[c] + 3 * h * 4, base + (size_t)(pw + mw + ew + 4) * 4, 0);
        }
      }
      size_t part = 0;
      for (OpenTask& t : tk) {
        const size_t need = open_task_plan(&t, B);
        t.partial = need ? w->open_partial + part : nullptr;
        part += need;
      }
      std::stable_sort(tk.begin(), tk.end(), [](const OpenTask& a, const OpenTask& b) { return a.kind < b.kind; });
      for (int k = 0; k < 6; ++k) w->open_first[k] = w->open_count[k] = w->open_blocks[k] = 0;
      w->open_cblocks = 0;
      for (size_t i = 0; i < tk.size(); ++i) {
        OpenTask& t = tk[i];
        if (w->open_count[t.kind]++ == 0) w->open_first[t.kind] = (int)i;
        t.blk0 = w->open_blocks[t.kind];
        w->open_blocks[t.kind] += t.blocks;
        t.cblk0 = w->open_cblocks;
        w->open_cblocks += t.cblocks;
      }
      if (tk.size() > (size_t)5 * kNumChips) return ctx->fail(3, "machine_prove: opening task table overflow");
      ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(w->open_tasks, tk.data(), tk.size() * sizeof(OpenTask), hipMemcpyHostToDevice, s));
      // the reduced openings' tables: heights from the tallest down, every matrix of every chip of a height
      w->mr_heights_host.clear(); w->mr_segs_host.clear(); w->mr_chips_host.clear();
      w->mr_blocks = 0;
      for (int lh = 31; lh >= 0; --lh) {
        MRHeight hh{};
        hh.logh = lh; hh.seg0 = (int)w->mr_segs_host.size(); hh.chip0 = (int)w->mr_chips_host.size(); hh.blk0 = w->mr_blocks;
        int first = -1;
        for (int c = 0; c < kNumChips; ++c) {
          if (logh[c] != lh) continue;
          if (first < 0) first = c;
          const ChipDef& d = chip_def(c);
          const Seg mats[4] = {prep_seg(c, true), rm[1].seg[c][0], rm[2].seg[c][0], rm[3].seg[c][0]};
          const int w0 = mats[0].width, n1 = w0 + mats[1].width + mats[2].width + mats[3].width, n2 = mats[1].width + mats[2].width;
          (void)d;
          int col0 = 0;
          for (int mi = 0; mi < 4; ++mi) {
            if (mats[mi].width) {
              MRSeg sg{};
              sg.p = mats[mi].p; sg.bstride = mats[mi].bstride; sg.width = mats[mi].width;
              sg.pow1 = (int)w->open_off[c] + col0;
              sg.pow2 = mi == 1 || mi == 2 ? (int)w->open_off[c] + n1 + col0 - w0 : -1;
              w->mr_segs_host.push_back(sg);
            }
            col0 += mats[mi].width;
          }
          w->mr_chips_host.push_back(MRChip{(int)w->open_off[c], n1, n2});
        }
        if (first < 0) continue;
        hh.nseg = (int)w->mr_segs_host.size() - hh.seg0;
        hh.nchips = (int)w->mr_chips_host.size() - hh.chip0;
        const size_t hgt = (size_t)1 << lh;
        if (lh == lm) { hh.out = w->fri_layers; hh.out_bstride = w->fri_layer_stride; }
        else { hh.out = w->G[lh]; hh.out_bstride = 2 * hgt * 4; }
        hh.tw_fwd = dom[first]->tw_fwd;
        hh.shift[0] = Fp::from_canonical(kGen).v;
        hh.shift[1] = (Fp::from_canonical(kGen) * fp_root_of_unity(lh + 1)).v;
        hh.w_h = dom[first]->w_h;
        w->mr_blocks += B * (int)((2 * hgt + kMReduceMultiPoints - 1) / kMReduceMultiPoints);
        w->mr_heights_host.push_back(hh);
      }
      ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(w->mr_heights, w->mr_heights_host.data(), w->mr_heights_host.size() * sizeof(MRHeight), hipMemcpyHostToDevice, s));
      ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(w->mr_segs, w->mr_segs_host.data(), w->mr_segs_host.size() * sizeof(MRSeg), hipMemcpyHostToDevice, s));
      ZKSP_HIP_CHECK(ctx, hipMemcpyAsync(w->mr_chips, w->mr_chips_host.data(), w->mr_chips_host.size() * sizeof(MRChip), hipMemcpyHostToDevice, s));
      w->open_tasks_batch = B;
    }
    if (tabled) {
      launch_open_multi(s, w->open_tasks, w->open_first, w->open_count, w->open_blocks, w->open_cblocks, 8 * R);
    } else {
      fork.begin();
      for (int c = 0; c < kNumChips; ++c) {
        const ChipDef& d = chip_def(c);
        const size_t h = H(c);
        const int pw = d.prep_w, mw = d.main_w, ew = d.perm_width();
        // Columns are opened from their evaluations (the traces, and the quotient values over their cosets) against
        // barycentric weights: table 0 at zeta, table 1 at zeta * w_H (table 0 moved by one place), tables 2 and 3 at zeta
        // for columns given on the cosets g<w_H> and g w_2H <w_H> (the two quotient chunks).  No coefficient arrays.
        const size_t zs = 4 * h * 4;
        uint32_t* const scratch = chip_scratch(c, w->reduce_scratch, w->side_reduce_scratch);
        uint32_t* base = w->opened + w->open_off[c] * 4;
        const size_t pt_stride = (size_t)mw + ew + (size_t)w->mat[c][2].w;
        // tall columns: split the rows over workgroups (the partial sums live in the reduce scratch, which is not in
        // use yet)
        auto open = [&](const uint32_t* evals, size_t cstride, int ncols, int npts, const uint32_t* table, uint32_t* dst, size_t pts) {
          if (logh[c] >= 12) launch_open_tall(SC(c), evals, cstride, ncols, logh[c], table, zs, npts, dst, 8 * R, pts, scratch, B);
          else launch_open(SC(c), evals, cstride, ncols, logh[c], table, zs, npts, dst, 8 * R, pts, B);
        };
        if (pw) open(prep->tr[PrepDevice::index_of(c)], 0, pw, 1, w->zpow[c], base, 0);
        open(w->mat[c][0].tr, (size_t)mw * h, mw, 2, w->zpow[c], base + (size_t)pw * 4, pt_stride);
        open(w->mat[c][1].tr, (size_t)ew * h, ew, 2, w->zpow[c], base + (size_t)(pw + mw) * 4, pt_stride);
        if (w->mat[c][2].w) {
          open(w->mat[c][2].tr, 8 * h, 4, 1, w->zpow[c] + 2 * h * 4, base + (size_t)(pw + mw + ew) * 4, 0);
          open(w->mat[c][2].tr + 4 * h, 8 * h, 4, 1, w->zpow[c] + 3 * h * 4, base + (size_t)(pw + mw + ew + 4) * 4, 0);
        }
      }
      fork.end();
    }
  }
  {
    ProfileSpan sp(ctx, "merkle_open");
    launch_merkle_commit(s, w->opened, 8 * R, 8, (int)w->open_rows_log, w->tree_o, (2 * R - 1) * 8, B, kc);
  }
  {
    ProfileSpan sp(ctx, "transcript");
    // alpha_f and delta; the coefficient of every opened value in the reduced openings (format v16)
    launch_ch_observe_sample(s, w->ch, w->tree_o + (2 * R - 2) * 8, (2 * R - 1) * 8, 8, w->af, 8, 2, B, kc);
    launch_reduce_coefs(s, w->af, 8, w->reduce_desc, w->af_pows, w->n_open * 4, (int)w->n_open, B);
  }
  // ---- reduced openings: one FRI input per height; the tallest is layer 0 ----
  {
    ProfileSpan sp(ctx, "m_reduce");
    bool seen[32] = {false};
    static const bool per_chip = getenv("ZKSP_REDUCE_PER_CHIP") != nullptr;  // measurement switch: the per-chip launches
    if (fork.lanes > 1 || !per_chip) {  // two launches over the tables built with the opening tasks
      MReduceMulti ma{};
      ma.heights = w->mr_heights; ma.n_heights = (int)w->mr_heights_host.size(); ma.segs = w->mr_segs; ma.chips = w->mr_chips;
      ma.af_pows = w->af_pows; ma.af_bstride = w->n_open * 4; ma.opened = w->opened; ma.opened_bstride = 8 * R; ma.zeta = w->zeta;
      ma.bsum = w->mr_bsum; ma.total_blocks = w->mr_blocks; ma.batch = B;
      launch_machine_reduce_multi(s, ma);
    } else {
    fork.begin();
    for (int c = 0; c < kNumChips; ++c) {
      const ChipDef& d = chip_def(c);
      const size_t h = H(c);
      MReduceArgs ra;
      ra.mats[0] = prep_seg(c, true);
      ra.mats[1] = rm[1].seg[c][0];
      ra.mats[2] = rm[2].seg[c][0];
      ra.mats[3] = rm[3].seg[c][0];
      ra.af_pows = w->af_pows;
      ra.af_bstride = w->n_open * 4;
      ra.pow_off = w->open_off[c];
      ra.opened = w->opened;
      ra.opened_bstride = 8 * R;
      ra.open_off = w->open_off[c];
      ra.zeta = w->zeta;
      ra.tw_fwd = dom[c]->tw_fwd;
      ra.shift[0] = g.v;
      ra.shift[1] = (g * fp_root_of_unity(logh[c] + 1)).v;
      ra.w_h = dom[c]->w_h;
      ra.partial = lane_scratch(c, w->reduce_scratch, w->side_reduce_scratch);
      ra.bsum = lane_scratch(c, w->bsum, w->side_bsum);
      if (logh[c] == lm) { ra.out = w->fri_layers; ra.out_bstride = w->fri_layer_stride; }
      else { ra.out = w->G[logh[c]]; ra.out_bstride = 2 * h * 4; }
      ra.accumulate = seen[logh[c]] ? 1 : 0;
      seen[logh[c]] = true;
      ra.logh = logh[c];
      ra.batch = B;
      (void)d;
      launch_machine_reduce(SL(c), ra);
    }
    fork.end();
    }
  }
  // ---- FRI commit phase; an input of height 2^k joins when the folded layer reaches that height ----
  size_t loff = 0, toff = 0;
  const size_t hmax = (size_t)1 << lm;
  const DeviceDomain* dmax = ctx->domain(lm);  // the tallest chip's domain (any chip may be the tallest)
  if (!dmax) return 3;
  // the layers of at most 2^kFriTailMaxLogLeaves leaves are committed, challenged and folded by ONE launch (fri_tail_kernel)
  const int k_tail = std::max(0, lm - kFriTailMaxLogLeaves);
  for (int k = 0; k < k_tail; ++k) {
    const int loghk = lm - k;
    const size_t hk = hmax >> k;
    {
      ProfileSpan sp(ctx, "fri_commit");
      launch_fri_commit(s, w->fri_layers + loff, w->fri_layer_stride, loghk, w->fri_trees + toff * 8, w->fri_tree_stride, B, kc);
    }
    {
      ProfileSpan sp(ctx, "transcript");
      launch_ch_observe_sample(s, w->ch, w->fri_trees + (toff + 2 * hk - 2) * 8, w->fri_tree_stride, 8, w->betas + (size_t)k * 4,
                               (size_t)lm * 4, 1, B, kc);
    }
    {
      ProfileSpan sp(ctx, "fri_fold");
      uint32_t* nxt = w->fri_layers + loff + 2 * hk * 4;
      launch_fri_fold(s, w->fri_layers + loff, w->fri_layer_stride, nxt, w->fri_layer_stride, w->betas + (size_t)k * 4,
                      (size_t)lm * 4, dmax->tw_inv, k, dmax->fold_xinv[2 * k], dmax->fold_xinv[2 * k + 1], loghk, B,
                      loghk - 1 >= 0 ? w->G[loghk - 1] : nullptr, hk * 4);  // (the input of that height joins in the same launch)
    }
    loff += 2 * hk * 4;
    toff += 2 * hk - 1;
  }
  if (k_tail < lm) {
    ProfileSpan sp(ctx, "fri_commit");
    FriTailArgs ta;
    ta.layers = w->fri_layers;
    ta.layer_stride = w->fri_layer_stride;
    ta.trees = w->fri_trees;
    ta.tree_stride = w->fri_tree_stride;
    ta.ch = w->ch;
    ta.betas = w->betas;
    ta.beta_stride = (size_t)lm * 4;
    ta.tw_inv = dmax->tw_inv;
    for (int i = 0; i < 48; ++i) ta.xinv[i] = i < 2 * lm ? dmax->fold_xinv[i] : 0;
    for (int k = 0; k < 24; ++k) ta.join[k] = k >= k_tail && k < lm && lm - k - 1 >= 0 ? w->G[lm - k - 1] : nullptr;
    ta.logh = lm;
    ta.k_start = k_tail;
    ta.loff_start = loff;
    ta.toff_start = toff;
    launch_fri_tail(s, ta, B, kc);
    for (int k = k_tail; k < lm; ++k) {
      const size_t hk = hmax >> k;
      loff += 2 * hk * 4;
      toff += 2 * hk - 1;
    }
  }
  {
    ProfileSpan sp(ctx, "transcript");
    launch_ch_observe_sample(s, w->ch, w->fri_layers + loff, w->fri_layer_stride, 4, w->alpha, 4, 0, B, kc);
  }
  {
    ProfileSpan sp(ctx, "grind");
    launch_ch_grind(s, w->ch, w->witness, pow_bits, B, kc, 1);
  }
  {
    ProfileSpan sp(ctx, "transcript");
    launch_ch_queries(s, w->ch, w->witness, w->indices, Q, pow_bits, lm + 1, B, kc, 1);
  }
  {
    ProfileSpan sp(ctx, "m_assemble");
    MAssembleArgs aa;
    memset(&aa, 0, sizeof aa);
    for (int r = 0; r < 4; ++r) {
      MRound& mr = aa.round[r];
      mr.lm = 0;
      for (int c = 0; c < kNumChips; ++c) {
        mr.logh[c] = logh[c];
        mr.seg[c] = rm[r].seg[c][0];
        if (mr.seg[c].width) mr.lm = std::max(mr.lm, logh[c]);
      }
      if (r == 0) { mr.tree = prep->tree; mr.tree_bstride = 0; }
      else { mr.tree = w->tree[r]; mr.tree_bstride = tree_stride; }
    }
    aa.cum = w->cum;
    aa.opened = w->opened;
    aa.opened_bstride = 8 * R;
    aa.n_open = w->n_open;
    aa.fri_layers = w->fri_layers;
    aa.fri_trees = w->fri_trees;
    aa.fri_layer_stride = w->fri_layer_stride;
    aa.fri_tree_stride = w->fri_tree_stride;
    aa.witness = w->witness;
    aa.indices = w->indices;
    aa.body = w->body;
    aa.body_stride = w->body_words;
    aa.lm = lm;
    aa.n_queries = Q;
    aa.batch = B;
    if (ctx->body_free) (void)hipStreamWaitEvent(s, ctx->body_free, 0);
    launch_machine_assemble(s, aa);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx->fail(3, std::string("machine_prove_resident: ") + hipGetErrorString(e));
  return 0;
}

}  // namespace zksp
